"""Every entry point of the C ABI called with NULL handles / pointers (run in a child process by tests/test_cabi.py: a crash would be a
segfault, not an exception).  Prints `name status` per call; no device is touched before the arguments are validated."""
import ctypes as C, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from signerf_amd import _lib
lib=_lib.load()
N=None
calls={
 "sn_destroy": lambda: lib.sn_destroy(N),
 "sn_last_error": lambda: lib.sn_last_error(N) is not None,
 "sn_create_null": lambda: lib.sn_create(None, None),
 "sn_upload_weights": lambda: lib.sn_upload_weights(N, b"x", N, 0, N),
 "sn_finalize_weights": lambda: lib.sn_finalize_weights(N, N),
 "sn_generate_rays": lambda: lib.sn_generate_rays(None, 1.0,1.0,0.0,0.0, 4,4, N,N,N,N, None, N,N, N),
 "sn_generate_rays_camera": lambda: lib.sn_generate_rays_camera(None, N, 0, N,N,N,N, None, N,N, N),
 "sn_intersect_with_aabb": lambda: lib.sn_intersect_with_aabb(N,N,4,None,N,N,N),
 "sn_intersect_obb": lambda: lib.sn_intersect_obb(N,N,4,None,None,N,N,N),
 "sn_workspace_bytes": lambda: lib.sn_workspace_bytes(N,4,4,None),
 "sn_render_rays": lambda: lib.sn_render_rays(N,N,N,N,N,4,4,None,N,N,N,N,N,N,N),
 "sn_render_rays_debug": lambda: lib.sn_render_rays_debug(N,N,N,N,N,4,4,None,N,N,N,N,N,N,None,N),
 "sn_render_normals": lambda: lib.sn_render_normals(N,N,N,N,N,4,4,None,N,N,N),
 "sn_effective_precision": lambda: lib.sn_effective_precision(N,1,0),
 "sn_debug_layout": lambda: lib.sn_debug_layout(N,-1,None),
 "sn_debug_read": lambda: lib.sn_debug_read(N,-1,0,N,0,N),
 "sn_debug_reload_env": lambda: lib.sn_debug_reload_env(N),
 "sn_debug_sample_positions": lambda: lib.sn_debug_sample_positions(N,N,N,N,4,N,N,N,N),
 "sn_clock_probe": lambda: lib.sn_clock_probe(N,0.01,N),
 "sn_hash_encode": lambda: lib.sn_hash_encode(N,-1,N,4,N,N,N),
 "sn_field_forward": lambda: lib.sn_field_forward(N,-1,N,N,4,1,N,N,N),
 "sn_field_forward_geo": lambda: lib.sn_field_forward_geo(N,-1,N,N,4,1,N,N,N,N),
 "sn_composite": lambda: lib.sn_composite(N,N,N,4,4,N,N,N,N,N,N,N),
 "sn_pdf_sample": lambda: lib.sn_pdf_sample(N,N,4,4,4,N,0.01,N,N,N),
 "sn_tensor_to_uint8": lambda: lib.sn_tensor_to_uint8(N,4,N,N),
 "sn_resize_bilinear": lambda: lib.sn_resize_bilinear(N,0,4,4,4,3,N,2,2,2,0,N),
 "sn_mask_workspace_bytes": lambda: lib.sn_mask_workspace_bytes(0,0),
 "sn_aabb_mask_condition": lambda: lib.sn_aabb_mask_condition(N,N,N,4,4,None,None,N,N,N,0,N),
}
only=sys.argv[1:] 
for k,f in calls.items():
    if only and k not in only: continue
    print(k, flush=True, end=' ')
    print(f(), flush=True)
