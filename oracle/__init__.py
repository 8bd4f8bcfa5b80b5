"""CPU oracle for the SIGNeRF reference-sheet render path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``signerf_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may, and there only as the checker / the CPU number that is
reported beside the GPU number -- never as the thing shipped or measured.

Pinning status
--------------
* ``oracle.signerf_utils``: restates in-tree reference code
  (``signerf/utils/intersection.py``, ``signerf/utils/poses_generation.py``,
  ``signerf/utils/image_tensor_converter.py``).  PINNED: checked bit-for-bit
  against fixtures under ``tests/golden/`` that were produced by importing the
  reference modules themselves (``tests/golden/make_golden.py``).
* ``oracle.nerfacto``: restates the eval-mode arithmetic of
  ``nerfstudio==1.0.2``'s pure-PyTorch ("torch" implementation) nerfacto path,
  which is what the reference reaches through
  ``signerf/datasetgenerator/datasetgenerator.py:691,694``.  nerfstudio is a
  third-party dependency (``pyproject.toml:6``) that is neither vendored in
  /root/reference nor installable here, and the reference ships no tests or
  golden vectors.  **PARITY UNPINNED** for this module: it follows SURVEY.md
  Appendix A (a recollection of nerfstudio 1.0.2) and is anchored only by the
  analytic known-answer tests in ``tests/test_oracle_kat.py``.
"""
