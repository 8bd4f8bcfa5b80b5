"""Stage-level operators: thin Python wrappers over the stage entry points of the C ABI (same device code as the fused
kernels).  Names follow the nerfstudio components they stand in for."""

from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def _f32(t: Tensor) -> Tensor:
    return t.to(torch.float32).contiguous()


def hash_encode(model, q: Tensor, which: int = -1, return_indices: bool = False):
    """HashEncoding forward (torch-path semantics, row a13).  q [n,3] in [0,1) -> features [n, L*F] (+ table rows [n,L,8])."""
    lib = model._ensure_engine()
    q = _f32(q)
    n = q.shape[0]
    enc = (model.field.mlp_base if which < 0 else model.proposal_networks[which].mlp_base).encoder
    with torch.cuda.device(q.device):
        feat = torch.empty((n, enc.num_levels * enc.features_per_level), dtype=torch.float32, device=q.device)
        idx = torch.empty((n, enc.num_levels, 8), dtype=torch.int32, device=q.device) if return_indices else None
        _lib.check(lib.sn_hash_encode(model._handle, which, _lib.ptr(q), n, _lib.ptr(feat), _lib.ptr(idx), _lib.current_stream()),
                   model._handle, "sn_hash_encode")
    return (feat, idx) if return_indices else feat


def field_forward(model, positions: Tensor, directions: Optional[Tensor] = None, which: int = -1, return_geo: bool = False):
    """NerfactoField / HashMLPDensityField on explicit world positions (rows a9, a14, a15) -> density [n], rgb [n,3] | None
    (+ the main field's geometry features [n,15] with ``return_geo``: nerfstudio's ``base_mlp_out``)."""
    lib = model._ensure_engine()
    pos = _f32(positions)
    n = pos.shape[0]
    d = None if directions is None else _f32(directions)
    with torch.cuda.device(pos.device):
        density = torch.empty((n,), dtype=torch.float32, device=pos.device)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=pos.device) if (which < 0 and d is not None) else None
        geo = torch.empty((n, 15), dtype=torch.float32, device=pos.device) if return_geo else None
        from .nerfacto import PRECISIONS

        model._engine_rw.acquire_read()
        try:
            _lib.check(lib.sn_field_forward_geo(model._handle, which, _lib.ptr(pos), _lib.ptr(d), n, PRECISIONS[model.config.precision],
                                                _lib.ptr(density), _lib.ptr(rgb), _lib.ptr(geo), _lib.current_stream()),
                       model._handle, "sn_field_forward")
        finally:
            model._engine_rw.release_read()
    return (density, rgb, geo) if return_geo else (density, rgb)


def sample_positions(origins: Tensor, directions: Tensor, starts: Tensor, ends: Tensor) -> dict:
    """Test instrumentation (sn_debug_sample_positions): the normalised, selector-multiplied positions q [n,3] of n explicit samples
    (origins / directions [n,3], starts / ends [n]) through the kernels' three position maps: ``strict`` (the literal torch-path arithmetic),
    ``exact`` (its division-free form in the main kernel behind the uniform sampler: must equal ``strict`` bit for bit) and ``fast``."""
    lib = _lib.load()
    o, d, t0, t1 = _f32(origins), _f32(directions), _f32(starts), _f32(ends)
    n = o.shape[0]
    with torch.cuda.device(o.device):
        out = {k: torch.empty((n, 3), dtype=torch.float32, device=o.device) for k in ("strict", "exact", "fast")}
        _lib.check(lib.sn_debug_sample_positions(_lib.ptr(o), _lib.ptr(d), _lib.ptr(t0), _lib.ptr(t1), n, _lib.ptr(out["strict"]),
                                                 _lib.ptr(out["exact"]), _lib.ptr(out["fast"]), _lib.current_stream()),
                   None, "sn_debug_sample_positions")
    return out


def composite(euclid_bins: Tensor, density: Tensor, rgb_samples: Tensor):
    """RaySamples.get_weights + RGB / median-depth / accumulation / expected-depth renderers (rows a10, a17).

    euclid_bins [R,S+1], density [R,S], rgb_samples [R,S,3] -> dict(weights [R,S], rgb [R,3], depth [R], median_index [R] int32,
    accumulation [R], expected_depth [R])."""
    lib = _lib.load()
    b, dn, c = _f32(euclid_bins), _f32(density), _f32(rgb_samples)
    R, S = dn.shape
    dev = b.device
    with torch.cuda.device(dev):
        out = {
            "weights": torch.empty((R, S), dtype=torch.float32, device=dev),
            "rgb": torch.empty((R, 3), dtype=torch.float32, device=dev),
            "depth": torch.empty((R,), dtype=torch.float32, device=dev),
            "median_index": torch.empty((R,), dtype=torch.int32, device=dev),
            "accumulation": torch.empty((R,), dtype=torch.float32, device=dev),
            "expected_depth": torch.empty((R,), dtype=torch.float32, device=dev),
        }
        _lib.check(lib.sn_composite(_lib.ptr(b), _lib.ptr(dn), _lib.ptr(c), R, S, _lib.ptr(out["weights"]), _lib.ptr(out["rgb"]),
                                    _lib.ptr(out["depth"]), _lib.ptr(out["median_index"]), _lib.ptr(out["accumulation"]),
                                    _lib.ptr(out["expected_depth"]), _lib.current_stream()), None, "sn_composite")
    return out


def pdf_sample(spacing_bins: Tensor, weights: Tensor, num_samples: int, histogram_padding: float = 0.01):
    """PDFSampler, eval mode (row a11): spacing_bins [R,N+1], weights [R,N] -> new bins [R,M+1], searchsorted indices [R,M+1]."""
    lib = _lib.load()
    sb, w = _f32(spacing_bins), _f32(weights)
    R, N = w.shape
    dev = sb.device
    nb = num_samples + 1
    u = (torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb) + 1.0 / (2 * nb)).to(dev)
    with torch.cuda.device(dev):
        bins = torch.empty((R, nb), dtype=torch.float32, device=dev)
        inds = torch.empty((R, nb), dtype=torch.int32, device=dev)
        _lib.check(lib.sn_pdf_sample(_lib.ptr(sb), _lib.ptr(w), R, N, num_samples, _lib.ptr(u), C.c_float(histogram_padding),
                                     _lib.ptr(bins), _lib.ptr(inds), _lib.current_stream()), None, "sn_pdf_sample")
    return bins, inds


def resize_bilinear(src: Tensor, out_h: int, out_w: int, threshold: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """``F.interpolate(src.permute(2,0,1)[None], (out_h, out_w), mode="bilinear", align_corners=False)`` back in [H,W,C] form
    (the reference's down/up-scale idiom, /root/reference/signerf/datasetgenerator/datasetgenerator.py:526-528,586,640-642,659).

    src: [H,W,C] fp32, or uint8/bool (the 0/1 mask: the reference's ``mask.float()``); may be a window of a larger image
    (``sheet[r0:r1, c0:c1, :]``).  threshold: return ``(value > 0.5)`` as 1.0/0.0 (``mask_scaled``).  out: optional
    [out_h,out_w,C] fp32 destination, may itself be a window of a sheet -- the resize then IS the paste (:537-539)."""
    lib = _lib.load()
    if src.dtype == torch.bool:
        src = src.view(torch.uint8) if src.is_contiguous() else src.to(torch.uint8)
    if src.dtype not in (torch.float32, torch.uint8):
        src = src.to(torch.float32)
    H, W, Cn = src.shape
    if src.stride(2) != 1 or src.stride(1) != Cn:
        src = src.contiguous()
    dev = src.device
    if out is None:
        with torch.cuda.device(dev):
            out = torch.empty((out_h, out_w, Cn), dtype=torch.float32, device=dev)
    if tuple(out.shape) != (out_h, out_w, Cn) or out.dtype != torch.float32 or out.stride(2) != 1 or out.stride(1) != Cn:
        raise _lib.SignerfHipError("resize_bilinear: `out` must be a channel-last fp32 [out_h, out_w, C] window")
    if not out.is_cuda or not src.is_cuda:
        raise _lib.SignerfHipError("resize_bilinear: tensors must live on the GPU (there is no CPU path)")
    with torch.cuda.device(dev):
        _lib.check(lib.sn_resize_bilinear(src.data_ptr(), int(src.dtype == torch.uint8), H, W, src.stride(0), Cn, out.data_ptr(), out_h, out_w,
                                          out.stride(0), int(bool(threshold)), _lib.current_stream()), None, "sn_resize_bilinear")
    return out


# ---- test instrumentation of the fused kernels (include/signerf_hip.h "SnDebugDump") -----------------------------------------
def render_rays_debug(model, ray_bundle, want=("main_fetch", "main_q", "median_index", "prop_fetch", "prop_q", "pdf_index")):
    """``Model.get_outputs_for_camera_ray_bundle`` through ``sn_render_rays_debug``: the production kernels instantiated with their
    DUMP flag.  Returns (outputs, dump): outputs = {"rgb","depth","accumulation","expected_depth"} [H,W,C]; dump holds
    "main_fetch" [H*W,S,16,8] int64 (uint32 words), "main_q" [H*W,S,3], "median_index" [H*W] int32, "prop_fetch_k" [H*W,N_k,5,8], "prop_q_k" [H*W,N_k,3],
    "pdf_index_k" [H*W,M_k+1] int32 -- see the header for the record format."""
    lib = model._ensure_engine()
    H, W = ray_bundle.origins.shape[:2]
    dev = model.device
    cfg = model.config
    f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    o, d, nears, fars = f32(ray_bundle.origins), f32(ray_bundle.directions), f32(ray_bundle.nears), f32(ray_bundle.fars)
    n = H * W
    S = cfg.num_nerf_samples_per_ray
    nprop = cfg.num_proposal_iterations
    with torch.cuda.device(dev):
        opts, keep = model._opts(H, W, lib)
        new = lambda c: torch.empty((n, c), dtype=torch.float32, device=dev)  # noqa: E731
        rgb, depth, acc, exp = new(3), new(1), new(1), new(1)
        dump = _lib.SnDebugDump()
        t = {}
        if "main_fetch" in want:
            t["main_fetch"] = torch.full((n, S, 16, 8), -1, dtype=torch.int32, device=dev)
            dump.main_fetch = t["main_fetch"].data_ptr()
        if "main_q" in want:
            t["main_q"] = torch.full((n, S, 3), float("nan"), dtype=torch.float32, device=dev)
            dump.main_q = t["main_q"].data_ptr()
        if "median_index" in want:
            t["median_index"] = torch.full((n,), -1, dtype=torch.int32, device=dev)
            dump.median_index = t["median_index"].data_ptr()
        counts = list(cfg.num_proposal_samples_per_ray[:nprop]) + [S]
        for k in range(nprop):
            if "prop_fetch" in want:
                t[f"prop_fetch_{k}"] = torch.full((n, counts[k], 5, 8), -1, dtype=torch.int32, device=dev)
                dump.prop_fetch[k] = t[f"prop_fetch_{k}"].data_ptr()
            if "prop_q" in want:
                t[f"prop_q_{k}"] = torch.full((n, counts[k], 3), float("nan"), dtype=torch.float32, device=dev)
                dump.prop_q[k] = t[f"prop_q_{k}"].data_ptr()
            if "pdf_index" in want:
                t[f"pdf_index_{k}"] = torch.full((n, counts[k + 1] + 1), -1, dtype=torch.int32, device=dev)
                dump.pdf_index[k] = t[f"pdf_index_{k}"].data_ptr()
        st = lib.sn_render_rays_debug(model._handle, _lib.ptr(o), _lib.ptr(d), _lib.ptr(nears), _lib.ptr(fars), H, W, C.byref(opts),
                                      _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(acc), _lib.ptr(exp), None, None, C.byref(dump),
                                      _lib.current_stream())
        _lib.check(st, model._handle, "sn_render_rays_debug")
        torch.cuda.synchronize(dev)
        del keep
    out = {"rgb": rgb.view(H, W, 3), "depth": depth.view(H, W, 1), "accumulation": acc.view(H, W, 1), "expected_depth": exp.view(H, W, 1)}
    for k in list(t):
        if "fetch" in k:  # uint32 words -> int64 so that the 0xA/0xD tags stay positive
            t[k] = t[k].to(torch.int64) & 0xFFFFFFFF
    return out, t


def debug_layout(model, which: int = -1) -> dict:
    """Layout of the derived gather buffers of one field (sn_debug_layout)."""
    lib = model._ensure_engine()
    lay = _lib.SnDebugLayout()
    _lib.check(lib.sn_debug_layout(model._handle, which, C.byref(lay)), model._handle, "sn_debug_layout")
    return {"n_dense": lay.n_dense, "n_bc": lay.n_bc, "dense_res": list(lay.dense_res), "dense_off": list(lay.dense_off),
            "dense_bytes": lay.dense_bytes, "pair_base": list(lay.pair_base), "pair_bytes": lay.pair_bytes, "feature_scale": lay.feature_scale,
            "table_bytes": lay.table_bytes, "handle_bytes": lay.handle_bytes, "half_grid_bytes": lay.half_grid_bytes}


def debug_read(model, which: int, what: int) -> Tensor:
    """Contents of the de-hashed copies (what = 0) or the x-paired tables (what = 1) as a flat fp32 tensor on the model's device."""
    lib = model._ensure_engine()
    lay = debug_layout(model, which)
    nbytes = lay["dense_bytes"] if what == 0 else lay["pair_bytes"]
    with torch.cuda.device(model.device):
        buf = torch.empty((nbytes // 4,), dtype=torch.float32, device=model.device)
        _lib.check(lib.sn_debug_read(model._handle, which, what, buf.data_ptr(), nbytes, _lib.current_stream()), model._handle, "sn_debug_read")
        torch.cuda.synchronize(model.device)
    return buf


def reload_env(model) -> None:
    """Has the handle re-read the SN_* diagnostic switches of the environment (they are otherwise read at sn_create /
    sn_finalize_weights only) -- for tests that flip one between two renders of the same model."""
    lib = model._ensure_engine()
    _lib.check(lib.sn_debug_reload_env(model._handle), model._handle, "sn_debug_reload_env")


def render_with_march_stats(model, ray_bundle):
    """``Model.get_outputs_for_camera_ray_bundle`` with ``SnRenderOpts.march_stats`` set: returns (outputs, {kernel: (wave-steps executed,
    wave-steps of the full march)}) for "K1" and, with proposal nets, "K2 level 0" / "K2 level 1".  executed < full only through the exact
    early termination of saturated waves.  A diagnostic: not thread-safe against other renders of the same model."""
    stats = torch.zeros(3, dtype=torch.int64, device=model.device)
    model._march_stats = stats
    try:
        out = model.get_outputs_for_camera_ray_bundle(ray_bundle)
        torch.cuda.synchronize(model.device)
    finally:
        model._march_stats = None
    skipped = [int(x) for x in stats.tolist()]
    H, W = ray_bundle.origins.shape[:2]
    tiles = ((W + 7) // 8) * ((H + 7) // 8) if H >= 8 else ((W + 63) // 64) * H   # the kernels' tile geometry (csrc/sn_api.hip tile_geometry)
    cfg = model.config
    full = tiles * cfg.num_nerf_samples_per_ray
    res = {"K1": (full - skipped[0], full)}
    for lv in range(cfg.num_proposal_iterations):
        full = tiles * cfg.num_proposal_samples_per_ray[lv]
        res[f"K2 level {lv}"] = (full - skipped[1 + lv], full)
    return out, res
