"""Builds libsignerf_hip.so (the C-ABI library) for gfx950 with hipcc, in-tree.

    python -m signerf_amd.build [--force]

The library is linked with ``-no-hip-rt`` against the HIP runtime PyTorch-ROCm ships
(torch/lib/libamdhip64.so) so that the process holds exactly ONE HIP runtime and raw device pointers
from torch tensors are valid inside the library.
"""

from __future__ import annotations

import os
import re
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libsignerf_hip.so")
SOURCES = ["sn_api.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "signerf_hip.h")]
ARCH = "gfx950"
# Code-generation flags of the device code (tools/isa_hazard_scan.py compiles with the same ones).
#   -fno-slp-vectorize, -disable-vector-combine: keep hipcc from packing neighbouring fp32 operations into v_pk_fma_f32 /
#   v_pk_add_f32 / v_pk_mul_f32.  Measured r02 (tools/probes/overlap2_probe.hip): on gfx950 packed-fp32 VALU instructions are
#   mutually exclusive with the matrix pipe of their SIMD, plain ones issue beside a running f16 MFMA.
CODEGEN_FLAGS = ["-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine"]


def _torch_lib_dir() -> str:
    import torch

    return os.path.join(os.path.dirname(torch.__file__), "lib")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=(), out_path: str = None) -> str:
    """out_path: build a VARIANT (extra_flags, e.g. ("-DSN_MFMA_PRIO=1",)) next to the product library, for A/B runs
    (SIGNERF_HIP_LIB=<out_path> selects it at load time)."""
    if out_path is None and not force and not _stale():
        return LIB_PATH
    target = out_path or LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tl = _torch_lib_dir()
    cmd = [hipcc, f"--offload-arch={ARCH}", *CODEGEN_FLAGS, "-fPIC", "-shared", "-no-hip-rt",
           "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Rpass-analysis=kernel-resource-usage",
           *extra_flags,
           *[os.path.join(CSRC, s) for s in SOURCES],
           "-o", target, f"-L{tl}", "-lamdhip64", f"-Wl,-rpath,{tl}", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    # keep real diagnostics, drop the resource-usage remarks and their source-context lines
    other, skip = [], 0
    for ln in r.stderr.splitlines():
        if "-Rpass-analysis" in ln:
            skip = 2  # the remark is followed by "  NNN | <source line>" and "      | ^"
            continue
        if skip and re.match(r"^\s*(\d+)?\s*\|", ln):
            skip -= 1
            continue
        skip = 0
        if ln.strip() and not ln.startswith("In file included from") and not re.match(r"^\d+ warnings? generated", ln):
            other.append(ln)
    if other:
        print("\n".join(other), file=sys.stderr)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    return target


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
