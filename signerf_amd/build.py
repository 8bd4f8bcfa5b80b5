"""Builds libsignerf_hip.so (the C-ABI library) for gfx950 with hipcc, in-tree.

    python -m signerf_amd.build [--force]

The library is linked with ``-no-hip-rt`` against the HIP runtime PyTorch-ROCm ships
(torch/lib/libamdhip64.so) so that the process holds exactly ONE HIP runtime and raw device pointers
from torch tensors are valid inside the library.
"""

from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libsignerf_hip.so")
SOURCES = ["sn_api.hip"]
HEADERS = ["sn_device.h", "sn_main.h", "sn_mask.h", "sn_proposal.h", "sn_stage.h", os.path.join("..", "..", "include", "signerf_hip.h")]
ARCH = "gfx950"


def _torch_lib_dir() -> str:
    import torch

    return os.path.join(os.path.dirname(torch.__file__), "lib")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tl = _torch_lib_dir()
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-no-hip-rt",
           "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
           *extra_flags,
           *[os.path.join(CSRC, s) for s in SOURCES],
           "-o", LIB_PATH, f"-L{tl}", "-lamdhip64", f"-Wl,-rpath,{tl}", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
