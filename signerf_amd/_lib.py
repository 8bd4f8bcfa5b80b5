"""ctypes binding of libsignerf_hip.so (the C ABI declared in include/signerf_hip.h).

There is NO CPU fallback: if the library is missing or a call fails this module raises.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

SN_MAX_LEVELS = 16
SN_MAX_PROPOSALS = 2
SN_OK, SN_ERR_INVALID, SN_ERR_HIP, SN_ERR_STATE, SN_ERR_WORKSPACE = 0, 1, 2, 3, 4
SN_ABI_VERSION = 6   # include/signerf_hip.h "ABI evolution": load() refuses a library built with another one

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# SIGNERF_HIP_LIB: load another build of the library (A/B experiments with tools/ab_lib.py); the default is the in-tree build
LIB_PATH = os.environ.get("SIGNERF_HIP_LIB") or os.path.join(_PKG_DIR, "libsignerf_hip.so")


class SnHashMlpDesc(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int32),
        ("features_per_level", C.c_int32),
        ("log2_hashmap_size", C.c_int32),
        ("hidden_dim", C.c_int32),
        ("num_layers", C.c_int32),
        ("out_dim", C.c_int32),
        ("scalings", C.c_float * SN_MAX_LEVELS),
        ("grid_mode", C.c_int32),
    ]


class _Sized(C.Structure):
    """A versioned struct of the C ABI: its first field is `struct_size` = sizeof(the struct) in THIS binding's layout, which the library uses
    to read (or write) no more than the binding knows (include/signerf_hip.h "ABI evolution").  Set at construction."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(self)


class SnFieldDesc(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("main_field", SnHashMlpDesc),
        ("geo_feat_dim", C.c_int32),
        ("hidden_dim_color", C.c_int32),
        ("appearance_embed_dim", C.c_int32),
        ("sh_levels", C.c_int32),
        ("sh_remap", C.c_int32),
        ("num_proposals", C.c_int32),
        ("proposals", SnHashMlpDesc * SN_MAX_PROPOSALS),
        ("average_init_density", C.c_float),
        ("histogram_padding", C.c_float),
        ("disable_scene_contraction", C.c_int32),
        ("aabb", C.c_float * 6),
        ("dense_levels", C.c_int32),
        ("dense_copy_cap_mb", C.c_int32),
        ("half_grid", C.c_int32),
    ]


class SnRenderOpts(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("num_proposal_iterations", C.c_int32),
        ("num_proposal_samples", C.c_int32 * SN_MAX_PROPOSALS),
        ("num_nerf_samples", C.c_int32),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
        ("chunk_rays", C.c_int32),
        ("precision", C.c_int32),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_size_t),
        ("initial_spacing_bins", C.c_void_p),
        ("pdf_u", C.c_void_p * SN_MAX_PROPOSALS),
        ("background_mode", C.c_int32),
        ("background_rgb", C.c_float * 3),
        ("spacing_mode", C.c_int32),
        ("march_stats", C.c_void_p),
        ("reuse_final_bins", C.c_int32),
    ]


class SnCameraDesc(C.Structure):
    _fields_ = [
        ("c2w", C.c_float * 12),
        ("fx", C.c_float),
        ("fy", C.c_float),
        ("cx", C.c_float),
        ("cy", C.c_float),
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("camera_type", C.c_int32),
        ("has_distortion", C.c_int32),
        ("distortion", C.c_float * 6),
    ]


class SnMaskOpts(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("inverse_mask", C.c_int32),
        ("dilate_w", C.c_int32),
        ("dilate_h", C.c_int32),
        ("has_manual_depth", C.c_int32),
        ("manual_min", C.c_double),
        ("manual_max", C.c_double),
        ("additional_depth_radius", C.c_float),
    ]


class SnDebugDump(C.Structure):
    _fields_ = [
        ("main_fetch", C.c_void_p),
        ("main_q", C.c_void_p),
        ("median_index", C.c_void_p),
        ("prop_fetch", C.c_void_p * SN_MAX_PROPOSALS),
        ("prop_q", C.c_void_p * SN_MAX_PROPOSALS),
        ("pdf_index", C.c_void_p * SN_MAX_PROPOSALS),
    ]


class SnDebugLayout(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_dense", C.c_int32),
        ("n_bc", C.c_int32),
        ("dense_res", C.c_uint32 * 12),
        ("dense_off", C.c_uint32 * 12),
        ("dense_bytes", C.c_uint64),
        ("pair_base", C.c_uint32 * SN_MAX_LEVELS),
        ("pair_bytes", C.c_uint64),
        ("feature_scale", C.c_float),
        ("table_bytes", C.c_uint64),
        ("handle_bytes", C.c_uint64),
        ("half_grid_bytes", C.c_uint64),
    ]


# name -> (restype, argtypes).  Must list every symbol include/signerf_hip.h declares
# (tests/test_cabi.py checks the two against each other).
_FP = C.c_void_p  # device pointer
SIGNATURES = {
    "sn_abi_version": (C.c_int, []),
    "sn_create": (C.c_int, [C.POINTER(SnFieldDesc), C.POINTER(C.c_void_p)]),
    "sn_destroy": (C.c_int, [C.c_void_p]),
    "sn_last_error": (C.c_char_p, [C.c_void_p]),
    "sn_upload_weights": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sn_finalize_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sn_generate_rays": (C.c_int, [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                   _FP, _FP, _FP, _FP, C.POINTER(C.c_float), _FP, _FP, C.c_void_p]),
    "sn_generate_rays_camera": (C.c_int, [C.POINTER(SnCameraDesc), _FP, C.c_int64, _FP, _FP, _FP, _FP, C.POINTER(C.c_float), _FP, _FP,
                                          C.c_void_p]),
    "sn_intersect_with_aabb": (C.c_int, [_FP, _FP, C.c_int64, C.POINTER(C.c_float), _FP, _FP, C.c_void_p]),
    "sn_intersect_obb": (C.c_int, [_FP, _FP, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float), _FP, _FP, C.c_void_p]),
    "sn_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SnRenderOpts)]),
    "sn_render_rays": (C.c_int, [C.c_void_p, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, C.POINTER(SnRenderOpts),
                                 _FP, _FP, _FP, _FP, _FP, _FP, C.c_void_p]),
    "sn_render_rays_debug": (C.c_int, [C.c_void_p, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, C.POINTER(SnRenderOpts),
                                       _FP, _FP, _FP, _FP, _FP, _FP, C.POINTER(SnDebugDump), C.c_void_p]),
    "sn_debug_layout": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(SnDebugLayout)]),
    "sn_debug_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _FP, C.c_size_t, C.c_void_p]),
    "sn_debug_reload_env": (C.c_int, [C.c_void_p]),
    "sn_debug_sample_positions": (C.c_int, [_FP, _FP, _FP, _FP, C.c_int64, _FP, _FP, _FP, C.c_void_p]),
    "sn_clock_probe": (C.c_int, [_FP, C.c_double, C.c_void_p]),
    "sn_effective_precision": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "sn_render_normals": (C.c_int, [C.c_void_p, _FP, _FP, _FP, _FP, C.c_int32, C.c_int32, C.POINTER(SnRenderOpts), _FP, _FP, C.c_void_p]),
    "sn_hash_encode": (C.c_int, [C.c_void_p, C.c_int32, _FP, C.c_int64, _FP, _FP, C.c_void_p]),
    "sn_field_forward": (C.c_int, [C.c_void_p, C.c_int32, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, C.c_void_p]),
    "sn_field_forward_geo": (C.c_int, [C.c_void_p, C.c_int32, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, _FP, C.c_void_p]),
    "sn_composite": (C.c_int, [_FP, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, C.c_void_p]),
    "sn_pdf_sample": (C.c_int, [_FP, _FP, C.c_int64, C.c_int32, C.c_int32, _FP, C.c_float, _FP, _FP, C.c_void_p]),
    "sn_tensor_to_uint8": (C.c_int, [_FP, C.c_int64, _FP, C.c_void_p]),
    "sn_resize_bilinear": (C.c_int, [_FP, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _FP, C.c_int32, C.c_int32, C.c_int64,
                                      C.c_int32, C.c_void_p]),
    "sn_mask_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sn_aabb_mask_condition": (C.c_int, [_FP, _FP, _FP, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(SnMaskOpts), _FP, _FP,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


class SignerfHipError(RuntimeError):
    """Raised when the HIP library is missing or one of its entry points reports an error."""


def load() -> C.CDLL:
    """Loads libsignerf_hip.so (once).  Imports torch first so the process shares ONE HIP runtime."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so before our library resolves it)

        if not os.path.exists(LIB_PATH):
            raise SignerfHipError(
                f"{LIB_PATH} not found: build it with `python -m signerf_amd.build` (there is no CPU fallback)")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise SignerfHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                raise SignerfHipError(f"{LIB_PATH} does not export {name}: it was built from another include/signerf_hip.h -- rebuild it "
                                      "(`python -m signerf_amd.build --force`)")
            fn.restype = res
            fn.argtypes = args
        got = lib.sn_abi_version()
        if got != SN_ABI_VERSION:
            raise SignerfHipError(f"{LIB_PATH} reports SN_ABI_VERSION {got}, this binding was written for {SN_ABI_VERSION}: rebuild the library")
        _lib = lib
        return lib


def check(status: int, handle=None, what: str = "") -> None:
    if status != 0:
        lib = load()
        msg = lib.sn_last_error(handle)
        raise SignerfHipError(f"{what} failed (status {status}): {msg.decode() if msg else ''}")


def ptr(t) -> Optional[int]:
    """Raw device pointer of a CUDA(HIP) fp32/int32 torch tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SignerfHipError("tensor must live on the GPU (there is no CPU path)")
    if not t.is_contiguous():
        raise SignerfHipError("tensor must be contiguous")
    return t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
