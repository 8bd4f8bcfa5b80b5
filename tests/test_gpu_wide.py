"""K1's wide launch shape (csrc/sn_main.h SnK1Shape: 8-wave workgroups of 4x2 tiles, four waves per SIMD, the tile-sequential split-precision
MLP ``sn_main_field_h4``, the waves' direction operands in LDS) against the 4-wave shape: the same MFMAs on the same operands in the same
order per accumulator, so every output must be BIT-IDENTICAL.  SN_K1_WIDE selects the shape per handle (read by ops.reload_env)."""
import pytest
import torch

from helpers import make_model
from signerf_amd import Cameras, SceneBox, ops, scene

pytestmark = pytest.mark.gpu

KEYS = ("rgb", "depth", "accumulation", "expected_depth")


def _both(model, bundle, monkeypatch, keys=KEYS):
    out = {}
    for wide in ("0", "1"):
        monkeypatch.setenv("SN_K1_WIDE", wide)
        ops.reload_env(model)
        out[wide] = {k: v.clone() for k, v in model.get_outputs_for_camera_ray_bundle(bundle).items() if k in keys}
    monkeypatch.delenv("SN_K1_WIDE")
    ops.reload_env(model)
    return out["0"], out["1"]


@pytest.mark.parametrize("H,W,cam", [(800, 800, 0), (808, 1000, 5), (731, 901, 2)])
def test_wide_shape_is_bit_identical_uniform_sampler(gpu, monkeypatch, H, W, cam):
    """BASELINE.json configs[1] at full size (and two frame sizes that are not multiples of the 32x16-pixel workgroup)."""
    cfg = scene.benchmark_config(64)
    model, _ = make_model(cfg, gpu)
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(gpu)
    a, b = _both(model, cams[cam].generate_rays(0), monkeypatch)
    for k in KEYS:
        assert torch.equal(a[k], b[k]), k
    assert float(a["rgb"].std()) > 0.05


def test_wide_shape_is_bit_identical_behind_the_proposal_sampler(gpu, monkeypatch):
    """configs[3] (1920x1080, 256 / 96 / 48) with a render box that some rays miss (per-ray nears / fars, NaN rays)."""
    cfg = scene.proposal_config()
    model, _ = make_model(cfg, gpu)
    H, W = 1080, 1920
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], 1.2 * H, 1.2 * H, W / 2, H / 2, W, H).to(gpu)
    keys = KEYS + ("prop_depth_0", "prop_depth_1")
    for box in (None, SceneBox(aabb=torch.tensor([[-0.3, -0.25, -0.2], [0.25, 0.3, 0.2]]))):
        a, b = _both(model, cams[3].generate_rays(0, aabb_box=box), monkeypatch, keys)
        for k in keys:
            assert torch.equal(a[k].nan_to_num(-7.0), b[k].nan_to_num(-7.0)), k


def test_wide_shape_tcnn_grid_and_small_frames(gpu, monkeypatch):
    """The tiny-cuda-nn grid instantiation; a frame below the threshold keeps the 4-wave shape either way (trivially identical, but it must render)."""
    cfg = scene.benchmark_config(32)
    cfg.implementation = "tcnn"
    model = cfg.setup().to(gpu).eval()
    W = H = 768
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(gpu)
    a, b = _both(model, cams[1].generate_rays(0), monkeypatch)
    for k in KEYS:
        assert torch.equal(a[k], b[k]), k
    small = Cameras(scene.benchmark_cameras(8)[:, :3], 96.0, 96.0, 48.0, 48.0, 96, 96).to(gpu)
    a, b = _both(model, small[1].generate_rays(0), monkeypatch)
    assert torch.equal(a["rgb"], b["rgb"])
