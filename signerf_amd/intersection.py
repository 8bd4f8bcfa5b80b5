"""``intersect_with_aabb`` on the GPU -- drop-in for /root/reference/signerf/utils/intersection.py:5-56 (row a4),
called on the full-resolution bundle right after the render (datasetgenerator.py:759-763)."""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def intersect_with_aabb(rays_o: torch.Tensor, rays_d: torch.Tensor, aabb: torch.Tensor):
    """rays_o, rays_d: [H,W,3] on the GPU; aabb: [2,3] -> nears, fars [H,W,1] (slab test, 1/(d+1e-6), no clamping)."""
    lib = _lib.load()
    H, W = rays_o.shape[0], rays_o.shape[1]
    o = rays_o.to(torch.float32).contiguous()
    d = rays_d.to(torch.float32).contiguous()
    box = (C.c_float * 6)(*aabb.detach().to("cpu", torch.float32).reshape(-1).tolist())
    with torch.cuda.device(o.device):
        nears = torch.empty((H, W, 1), dtype=torch.float32, device=o.device)
        fars = torch.empty((H, W, 1), dtype=torch.float32, device=o.device)
        _lib.check(lib.sn_intersect_with_aabb(_lib.ptr(o), _lib.ptr(d), H * W, box, _lib.ptr(nears), _lib.ptr(fars),
                                              _lib.current_stream()), None, "sn_intersect_with_aabb")
    return nears, fars
