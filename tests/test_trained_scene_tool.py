"""CPU checks of the test infrastructure the trained-scene legs stand on (tools/make_trained_scene.py, the oracle's sensitivity
probes): nothing here touches the HIP library."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from helpers import oracle_config, small_config  # noqa: E402
from oracle import nerfacto as onf  # noqa: E402


def _digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def test_the_fit_is_run_to_run_identical_and_leaves_the_global_switch_alone():
    import make_trained_scene as mts

    cfg = small_config()
    was = torch.are_deterministic_algorithms_enabled()
    a, meta = mts.fit(cfg, "cpu", steps=3, points=2048)
    b, _ = mts.fit(cfg, "cpu", steps=3, points=2048)
    assert _digest(a) == _digest(b) and meta["steps"] == 3
    assert torch.are_deterministic_algorithms_enabled() == was
    init = mts.initial_state_dict(cfg, 0)
    moved = [k for k in a if k.endswith("hash_table") and not torch.equal(a[k], init[k])]
    assert len(moved) == 1 + cfg.num_proposal_iterations      # every table was fitted


def test_exp_modes_differ_from_torch_exp_by_alpha_quanta_only():
    g = torch.Generator().manual_seed(3)
    deltas = torch.rand(64, 32, 1, generator=g) * 1e-2
    dens = torch.exp(torch.randn(64, 32, 1, generator=g) * 4.0 - 6.0)
    w0 = onf.get_weights(deltas, dens)
    for mode in ("rounded", "exp2"):
        onf.EXP_MODE = mode
        try:
            w = onf.get_weights(deltas, dens)
        finally:
            onf.EXP_MODE = "torch"
        d = (w - w0).abs()
        assert 0 < int((d > 0).sum()) < d.numel() // 2          # some last bits differ ...
        assert float(d.max()) <= 4 * 2.0 ** -24                   # ... by quanta of an alpha (T <= 1)
    assert onf.EXP_MODE == "torch"


def test_weight_nudge_reaches_the_resampler_only():
    cfg = small_config(num_proposal_samples_per_ray=(8, 4), num_nerf_samples_per_ray=4)
    from signerf_amd import scene

    sd = scene.synthetic_state_dict(cfg, seed=1, density_bias=3.0)
    ocfg = oracle_config(cfg)
    o = torch.tensor([[0.3, -0.2, 0.4]])
    d = torch.nn.functional.normalize(torch.tensor([[-0.5, 0.3, -0.8]]), dim=-1)
    with torch.no_grad():
        base = onf.get_outputs(sd, ocfg, o, d)
        zero = onf.get_outputs(sd, ocfg, o, d, weight_nudge={0: torch.zeros(1, 8, 1)})
        big = torch.zeros(1, 8, 1)
        big[0, 2, 0] = 0.5
        moved = onf.get_outputs(sd, ocfg, o, d, weight_nudge={0: big}, return_debug=True)
    assert all(torch.equal(base[k], zero[k]) for k in ("rgb", "depth", "accumulation"))
    assert torch.equal(moved["prop_depth_0"], base["prop_depth_0"])          # level 0's own outputs are taken before the nudge
    assert not torch.equal(moved["_debug"]["pdf_inds_1"], onf.get_outputs(sd, ocfg, o, d, return_debug=True)["_debug"]["pdf_inds_1"])
