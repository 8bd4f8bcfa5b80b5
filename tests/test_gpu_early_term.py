"""Exact early termination of saturated waves (csrc/sn_main.h, csrc/sn_proposal.h; r04): once exp(-cumsum(tau)) has underflowed to exactly 0 for all 64 rays of a
wave, every later weight is exactly +0, so the march may stop (K2) or jump to the last sample (K1: its colour is the 'last_sample' background) without changing any
output bit.  SN_EARLY_TERM=0 evaluates every sample: the two renders must be BIT-IDENTICAL on scenes that do saturate (dense media), that partly saturate, and that
never do (the benchmark scene), with render boxes (rays that miss carry NaN sums and never count as saturated), every background and both grids."""
import time

import pytest
import torch

from helpers import make_model, small_config, synthetic_tcnn_checkpoint
from signerf_amd import Cameras, SceneBox, ops, scene

pytestmark = pytest.mark.gpu

KEYS = ("rgb", "depth", "accumulation", "expected_depth", "prop_depth_0", "prop_depth_1")


def _both(model, bundle, monkeypatch):
    out = {}
    for et in ("0", "1"):
        monkeypatch.setenv("SN_EARLY_TERM", et)
        ops.reload_env(model)
        o = model.get_outputs_for_camera_ray_bundle(bundle)
        out[et] = {k: o[k].clone() for k in KEYS if k in o}
    monkeypatch.delenv("SN_EARLY_TERM")
    ops.reload_env(model)
    return out["0"], out["1"]


def _same(a, b):
    for k in a:
        assert torch.equal(a[k].nan_to_num(-7.0), b[k].nan_to_num(-7.0)), k
        assert torch.equal(torch.isnan(a[k]), torch.isnan(b[k])), k


@pytest.mark.parametrize("bias", [4.0, 9.0, 14.0])          # the benchmark's medium (never saturates), a dense one, an opaque one
@pytest.mark.parametrize("props", [0, 2])
@pytest.mark.parametrize("background", ["last_sample", "white"])
def test_early_termination_is_bit_identical(gpu, monkeypatch, bias, props, background):
    kw = dict(num_proposal_iterations=props, background_color=background)
    if props:
        kw.update(num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=24)
    else:
        kw.update(num_nerf_samples_per_ray=48)
    cfg = small_config(**kw)
    model, _ = make_model(cfg, gpu, density_bias=bias)
    H, W = 72, 104
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], 1.1 * H, 1.1 * H, W / 2, H / 2, W, H).to(gpu)
    for cam, box in ((0, None), (3, None), (5, SceneBox(aabb=torch.tensor([[-0.3, -0.25, -0.2], [0.25, 0.3, 0.2]])))):
        model.render_aabb = box
        a, b = _both(model, cams[cam].generate_rays(0, aabb_box=box), monkeypatch)
        _same(a, b)
        if bias >= 9.0 and box is None:
            assert float(a["accumulation"].mean()) > 0.9     # the medium is dense: the termination had work to skip


def test_early_termination_tcnn_grid_and_fp32(gpu, monkeypatch):
    cfg = small_config(implementation="tcnn", average_init_density=400.0, num_proposal_iterations=2, num_proposal_samples_per_ray=(48, 24),
                       num_nerf_samples_per_ray=16, precision="fp32")
    model = cfg.setup()
    model.load_state_dict(synthetic_tcnn_checkpoint(cfg, seed=1), strict=False)
    model = model.to(gpu).eval()
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], 70.0, 70.0, 28.0, 20.0, 56, 40).to(gpu)
    a, b = _both(model, cams[2].generate_rays(0), monkeypatch)
    _same(a, b)


def test_early_termination_pays_on_an_opaque_scene(gpu, monkeypatch):
    """Not a parity test: on a medium that saturates within the first samples the full-size launch gets much shorter (a trained scene's
    rays saturate a few samples behind the surface they hit); on the benchmark's medium it must not cost anything measurable."""
    def ms(model, bundle, et):
        monkeypatch.setenv("SN_EARLY_TERM", et)
        ops.reload_env(model)
        for _ in range(3):
            model.get_outputs_for_camera_ray_bundle(bundle)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            model.get_outputs_for_camera_ray_bundle(bundle)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 100.0

    W = H = 640
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(gpu)
    res = {}
    for name, cfg, bias in (("opaque, uniform sampler", scene.benchmark_config(64), 16.0), ("opaque, proposal sampler", scene.proposal_config(), 16.0),
                            ("benchmark medium", scene.benchmark_config(64), 4.0)):
        model, _ = make_model(cfg, gpu, density_bias=bias)
        b = cams[0].generate_rays(0)
        res[name] = (ms(model, b, "0"), ms(model, b, "1"))
        print(f"early termination, 640x640, {name}: {res[name][0]:.3f} -> {res[name][1]:.3f} ms per frame")
    monkeypatch.delenv("SN_EARLY_TERM")
    assert res["opaque, uniform sampler"][1] < 0.8 * res["opaque, uniform sampler"][0]
    assert res["opaque, proposal sampler"][1] < 0.9 * res["opaque, proposal sampler"][0]
    assert res["benchmark medium"][1] < 1.03 * res["benchmark medium"][0]


@pytest.mark.parametrize("colour_bias", [-12.0, -25.0, 6.0])
def test_early_termination_with_near_black_and_near_white_colours(gpu, monkeypatch, colour_bias):
    """Scenes whose colours are ~1e-5 / ~1e-11 (sigmoid of a strongly negative last-layer bias: a black object) or ~1: the colour sums are
    the smallest / largest accumulators a ray holds.  The termination (T == 0 for the whole wave) must be bit-identical to the full march in all
    of them, dense and opaque media, with and without a render box.  (r05 built and measured an EARLIER exact exit that hangs on exactly these
    accumulators -- T 2^26 below every fp32 sum of the ray -- and rejected it on speed: tools/patches/r05_k1_earlier_exact_exit.patch; this test
    is what showed it bit-identical.)"""
    cfg = small_config(num_proposal_iterations=2, num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=32)
    for density_bias in (9.0, 14.0):
        model, sd = make_model(cfg, gpu, density_bias=density_bias)
        sd["field.mlp_head.layers.2.bias"] = sd["field.mlp_head.layers.2.bias"] + colour_bias
        model.load_state_dict(sd, strict=False)
        model = model.to(gpu).eval()
        H, W = 80, 96
        cams = Cameras(scene.benchmark_cameras(8)[:, :3], 1.1 * H, 1.1 * H, W / 2, H / 2, W, H).to(gpu)
        for cam, box in ((1, None), (6, SceneBox(aabb=torch.tensor([[-0.3, -0.25, -0.2], [0.25, 0.3, 0.2]])))):
            model.render_aabb = box
            a, b = _both(model, cams[cam].generate_rays(0, aabb_box=box), monkeypatch)
            _same(a, b)
            if box is None:
                assert float(a["accumulation"].mean()) > 0.9
                print(f"colour bias {colour_bias:+.0f}, density bias {density_bias:+.0f}: rgb mean {float(a['rgb'].mean()):.3e}")
