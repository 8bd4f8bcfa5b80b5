"""The C ABI's evolution guards on the GPU (include/signerf_hip.h "ABI evolution", r06), through raw ctypes calls:
  * a caller compiled against a SHORTER SnRenderOpts / SnDebugLayout: the library neither reads nor writes behind the size the caller declares
    (a wild `march_stats` pointer in the memory behind a short SnRenderOpts is never written through);
  * SnRenderOpts.reuse_final_bins: the host-side stamp of a workspace's final bins -- sn_render_normals returns SN_ERR_STATE when the bins
    in the workspace are not this call's (another bundle, frame size, handle or weights epoch; a workspace no render ever wrote; one the
    mask step used since) and renders bit-identically to the stand-alone launch when they are."""
import ctypes as C

import pytest
import torch

from helpers import make_model, small_config
from signerf_amd import Cameras, _lib, scene

pytestmark = pytest.mark.gpu


def _setup(gpu, H=40, W=48, seed=0):
    cfg = small_config(num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=12, predict_normals=True)
    model, _ = make_model(cfg, gpu, seed=seed)
    model.eval()
    lib = model._ensure_engine()
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(gpu)
    return cfg, model, lib, cams


def _rays(cams, i):
    b = cams[i].generate_rays(camera_indices=0)
    return b.origins.reshape(-1, 3).contiguous(), b.directions.reshape(-1, 3).contiguous()


def _render(model, lib, o, d, H, W, opts):
    n = H * W
    out = {k: torch.empty((n, c), dtype=torch.float32, device=o.device) for k, c in (("rgb", 3), ("depth", 1), ("acc", 1), ("exp", 1), ("p0", 1), ("p1", 1))}
    st = lib.sn_render_rays(model._handle, _lib.ptr(o), _lib.ptr(d), None, None, H, W, C.byref(opts), _lib.ptr(out["rgb"]), _lib.ptr(out["depth"]),
                            _lib.ptr(out["acc"]), _lib.ptr(out["exp"]), _lib.ptr(out["p0"]), _lib.ptr(out["p1"]), _lib.current_stream())
    return st, out


def _normals(model, lib, o, d, H, W, opts, reuse):
    nrm = torch.full((H * W, 3), -5.0, dtype=torch.float32, device=o.device)
    opts.reuse_final_bins = 1 if reuse else 0
    st = lib.sn_render_normals(model._handle, _lib.ptr(o), _lib.ptr(d), None, None, H, W, C.byref(opts), _lib.ptr(nrm), None, _lib.current_stream())
    msg = (lib.sn_last_error(model._handle) or b"").decode()
    return st, nrm, msg


def test_short_render_opts_is_never_read_or_written_behind_its_declared_size(gpu):
    H, W = 40, 48
    cfg, model, lib, cams = _setup(gpu, H, W)
    o, d = _rays(cams, 0)
    full, keep = model._opts(H, W, lib)
    st, ref = _render(model, lib, o, d, H, W, full)
    assert st == 0
    short, keep2 = model._opts(H, W, lib)
    short.struct_size = _lib.SnRenderOpts.march_stats.offset      # a binding that predates march_stats / reuse_final_bins
    short.march_stats = 0xDEAD0000                                  # what its memory holds THERE is not a pointer the library may follow
    short.reuse_final_bins = 0x7FFFFFFF
    assert lib.sn_workspace_bytes(model._handle, H, W, C.byref(short)) == lib.sn_workspace_bytes(model._handle, H, W, C.byref(full)) > 0
    st, got = _render(model, lib, o, d, H, W, short)
    torch.cuda.synchronize()
    assert st == 0
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    assert short.struct_size == _lib.SnRenderOpts.march_stats.offset and short.march_stats == 0xDEAD0000     # the caller's struct is const
    # sizes the library does not know
    for bad, word in ((0, "was not set"), (C.sizeof(_lib.SnRenderOpts) + 8, "newer header"), (_lib.SnRenderOpts.march_stats.offset - 4, "knows sizes")):
        bad_o, _ = model._opts(H, W, lib)
        bad_o.struct_size = bad
        st, _ = _render(model, lib, o, d, H, W, bad_o)
        msg = (lib.sn_last_error(model._handle) or b"").decode()
        assert st == _lib.SN_ERR_INVALID and "struct_size" in msg and word in msg, msg
        assert lib.sn_workspace_bytes(model._handle, H, W, C.byref(bad_o)) == 0


def test_short_debug_layout_is_not_overrun(gpu):
    cfg, model, lib, cams = _setup(gpu)

    class Padded(C.Structure):      # the caller's (short) struct followed by memory that is NOT the library's to write
        _fields_ = [("lay", _lib.SnDebugLayout), ("canary", C.c_uint64 * 4)]

    p = Padded()
    short = _lib.SnDebugLayout.table_bytes.offset
    p.lay.struct_size = short
    p.lay.table_bytes = p.lay.handle_bytes = p.lay.half_grid_bytes = 0x1111111111111111
    for i in range(4):
        p.canary[i] = 0x2222222222222222
    assert lib.sn_debug_layout(model._handle, -1, C.byref(p.lay)) == 0
    assert p.lay.n_dense > 0 and p.lay.feature_scale > 0 and p.lay.struct_size == short
    assert p.lay.table_bytes == p.lay.handle_bytes == p.lay.half_grid_bytes == 0x1111111111111111     # behind the declared size: untouched
    assert all(p.canary[i] == 0x2222222222222222 for i in range(4))
    full = _lib.SnDebugLayout()
    assert lib.sn_debug_layout(model._handle, -1, C.byref(full)) == 0 and full.table_bytes > 0 and full.handle_bytes >= full.table_bytes
    full.struct_size = 0
    assert lib.sn_debug_layout(model._handle, -1, C.byref(full)) == _lib.SN_ERR_INVALID


def test_reuse_final_bins_is_refused_on_a_workspace_that_does_not_hold_this_frames_bins(gpu):
    H, W = 40, 48
    cfg, model, lib, cams = _setup(gpu, H, W)
    oX, dX = _rays(cams, 0)
    oY, dY = _rays(cams, 3)
    opts, keep = model._opts(H, W, lib)

    # the stand-alone launch (runs the proposal kernel itself) is the reference
    st, want, _ = _normals(model, lib, oX, dX, H, W, opts, reuse=False)
    assert st == 0
    # ... and leaves the bins of bundle X in the workspace: a reuse launch right behind it is legitimate and bit-identical
    st, got, _ = _normals(model, lib, oX, dX, H, W, opts, reuse=True)
    assert st == 0 and torch.equal(want, got)

    # colour render of X, then normals with reuse: the production sequence
    assert _render(model, lib, oX, dX, H, W, opts)[0] == 0
    st, got, _ = _normals(model, lib, oX, dX, H, W, opts, reuse=True)
    assert st == 0 and torch.equal(want, got)

    # another bundle rendered into the SAME workspace since: X's bins are gone
    assert _render(model, lib, oY, dY, H, W, opts)[0] == 0
    st, got, msg = _normals(model, lib, oX, dX, H, W, opts, reuse=True)
    assert st == _lib.SN_ERR_STATE and "another ray bundle" in msg and float(got.min()) == -5.0 == float(got.max())     # nothing was launched

    # a workspace no render ever wrote
    fresh, keep_f = model._opts(H, W, lib)
    st, _, msg = _normals(model, lib, oX, dX, H, W, fresh, reuse=True)
    assert st == _lib.SN_ERR_STATE and "no sn_render_rays call" in msg

    # same workspace, other sample counts / frame size
    assert _render(model, lib, oX, dX, H, W, opts)[0] == 0
    opts.num_nerf_samples = 8
    st, _, msg = _normals(model, lib, oX, dX, H, W, opts, reuse=True)
    assert st == _lib.SN_ERR_STATE and "sample counts" in msg
    opts.num_nerf_samples = cfg.num_nerf_samples_per_ray
    st, _, msg = _normals(model, lib, oX[: (H - 8) * W], dX[: (H - 8) * W], H - 8, W, opts, reuse=True)
    assert st == _lib.SN_ERR_STATE and "another size" in msg

    # the weights changed in between (any upload advances the handle's epoch)
    assert _render(model, lib, oX, dX, H, W, opts)[0] == 0
    mean = model.field.embedding_appearance.mean(0).detach().to(torch.float32).contiguous()
    _lib.check(lib.sn_upload_weights(model._handle, b"field.embedding_appearance.mean", _lib.ptr(mean), mean.numel() * 4, _lib.current_stream()), model._handle, "upload")
    _lib.check(lib.sn_finalize_weights(model._handle, _lib.current_stream()), model._handle, "finalize")
    st, _, msg = _normals(model, lib, oX, dX, H, W, opts, reuse=True)
    assert st == _lib.SN_ERR_STATE and "weights changed" in msg
    st, got, _ = _normals(model, lib, oX, dX, H, W, opts, reuse=False)          # the stand-alone launch still works, same picture (same values uploaded)
    assert st == 0 and torch.equal(want, got)

    # the mask step was handed the workspace since
    assert _render(model, lib, oX, dX, H, W, opts)[0] == 0
    depth = torch.ones((H * W,), dtype=torch.float32, device=gpu)
    mask = torch.empty((H * W,), dtype=torch.uint8, device=gpu)
    aabb = (C.c_float * 6)(-0.1, -0.1, -0.1, 0.1, 0.1, 0.1)
    mo = _lib.SnMaskOpts()
    mo.additional_depth_radius = 0.1
    assert keep[0].numel() >= lib.sn_mask_workspace_bytes(H, W)
    st = lib.sn_aabb_mask_condition(_lib.ptr(oX), _lib.ptr(dX), _lib.ptr(depth), H, W, aabb, C.byref(mo), _lib.ptr(mask), None, keep[0].data_ptr(),
                                    keep[0].numel(), _lib.current_stream())
    assert st == 0
    st, _, msg = _normals(model, lib, oX, dX, H, W, opts, reuse=True)
    assert st == _lib.SN_ERR_STATE and "no sn_render_rays call" in msg
    # a mask-opts struct whose size the library does not know
    mo.struct_size = 4
    assert lib.sn_aabb_mask_condition(_lib.ptr(oX), _lib.ptr(dX), _lib.ptr(depth), H, W, aabb, C.byref(mo), _lib.ptr(mask), None, keep[0].data_ptr(),
                                      keep[0].numel(), _lib.current_stream()) == _lib.SN_ERR_INVALID


def test_reuse_final_bins_is_refused_across_handles(gpu):
    H, W = 40, 48
    cfg, model_a, lib, cams = _setup(gpu, H, W)
    _, model_b, _, _ = _setup(gpu, H, W, seed=1)
    o, d = _rays(cams, 1)
    opts, keep = model_a._opts(H, W, lib)
    assert _render(model_a, lib, o, d, H, W, opts)[0] == 0
    st, _, msg = _normals(model_b, lib, o, d, H, W, opts, reuse=True)      # B is asked to march the bins A's proposal nets made
    assert st == _lib.SN_ERR_STATE and "another handle" in msg
    assert _render(model_b, lib, o, d, H, W, opts)[0] == 0                  # B renders into the workspace: now they are B's ...
    st, _, msg = _normals(model_a, lib, o, d, H, W, opts, reuse=True)      # ... and no longer A's
    assert st == _lib.SN_ERR_STATE and "another handle" in msg
    assert _normals(model_b, lib, o, d, H, W, opts, reuse=True)[0] == 0


def test_model_level_lazy_normals_survive_a_bundle_that_needed_a_copy(gpu):
    """The shim hands the normals launch the very tensors the colour render marched: a bundle that is not fp32 / contiguous (so that the
    render works on a converted COPY) still re-uses its bins -- no warning, same normals as compute_normals="always"."""
    import warnings

    H, W = 40, 48
    cfg, model, lib, cams = _setup(gpu, H, W)
    b = cams[2].generate_rays(camera_indices=0)
    b64 = b._map(lambda t: t.double() if t.is_floating_point() else t)      # fp64 rays: every render call converts them
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = model.get_outputs_for_camera_ray_bundle(b64)
        lazy = out["normals"].clone()
    model.config.compute_normals = "always"
    always = model.get_outputs_for_camera_ray_bundle(b)["normals"]
    model.config.compute_normals = "lazy"
    assert torch.equal(lazy, always)
    # lazy reuse can be capped by memory: with a cap of 0 MB the state is not kept and the normals launch runs the proposal kernel itself
    model.config.normals_bin_reuse_max_mb = 0
    out = model.get_outputs_for_camera_ray_bundle(b)
    assert torch.equal(out["normals"], always)
