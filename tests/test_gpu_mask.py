"""SURVEY §8(f) row 1 on the GPU: mask + condition step of render_camera (aabb mode) against the oracle's restatement of
datasetgenerator.py:758-818.  The mask is boolean work -> bit-exact; the condition is strict IEEE arithmetic -> bit-exact."""
import pytest
import torch

from helpers import make_model, small_config
from oracle import nerfacto as onf
from oracle import signerf_utils as su
from signerf_amd import Cameras, scene
from signerf_amd.datasetgenerator import DatasetGeneratorConfig, aabb_mask_and_condition, render_camera

pytestmark = pytest.mark.gpu


def _scene(H, W, cam=0, focal=None):
    c2w = scene.benchmark_cameras(8)
    r = onf.generate_rays(c2w[cam, :3], focal or float(W), focal or float(W), W / 2, H / 2, H, W)
    g = torch.Generator().manual_seed(H * W + cam)
    # far background with one patch (plus a few isolated pixels) whose depth falls inside the box
    depth = 2.0 + torch.rand(H, W, 1, generator=g)
    y0, y1, x0, x1 = (3 * H) // 8, (5 * H) // 8, (3 * W) // 8, (5 * W) // 8
    depth[y0:y1, x0:x1] = 0.45 + 0.1 * torch.rand(y1 - y0, x1 - x0, 1, generator=g)
    depth[H // 2, 1] = 0.5
    depth[1, W // 2] = 0.5
    return r["origins"], r["directions"], depth


@pytest.mark.parametrize("H,W,dil,inverse,manual", [
    (120, 160, (50, 50), False, None),       # the reference's defaults
    (97, 131, (9, 5), False, None),          # ragged size, non-square element
    (64, 64, None, False, None),             # no dilation
    (80, 100, (50, 50), True, None),         # inverse mask
    (80, 100, (21, 21), False, (0.1, 0.9)),  # manual depth range
    (40, 40, (64, 64), False, None),         # element larger than the image
])
def test_mask_and_condition_match_oracle(gpu, H, W, dil, inverse, manual):
    o, d, depth = _scene(H, W, cam=1, focal=1.4 * W)
    aabb = torch.tensor([[-0.1, -0.1, -0.1], [0.1, 0.1, 0.1]])
    mask, cond = aabb_mask_and_condition(depth.to(gpu), o.to(gpu), d.to(gpu), aabb, dil, inverse, manual, 0.1)
    rmask, rcond = su.aabb_mask_and_condition(depth, o, d, aabb, dil, inverse, manual, 0.1)
    assert mask.dtype == torch.bool and mask.shape == (H, W, 1) and cond.shape == (H, W, 1)
    rvis, _ = su.aabb_mask_and_condition(depth, o, d, aabb, None, False, manual, 0.1)
    assert 0.005 < float(rvis.float().mean()) < 0.5             # non-vacuous: some rays see the box, most do not
    if dil is not None and not inverse:
        assert int(rmask.sum()) > int(rvis.sum())               # the dilation really grows the mask
    assert torch.equal(mask.cpu(), rmask)                      # BIT-EXACT mask (slab test, dilation footprint, border handling)
    assert torch.equal(cond.cpu(), rcond)                      # strict IEEE: (depth - dmin) / (dmax - dmin), 1 - clamp


def test_nothing_visible_gives_zero_mask_and_condition(gpu):
    o, d, depth = _scene(48, 48)
    aabb = torch.tensor([[5.0, 5.0, 5.0], [5.1, 5.1, 5.1]])  # box nowhere near the rays
    mask, cond = aabb_mask_and_condition(depth.to(gpu), o.to(gpu), d.to(gpu), aabb)
    rmask, rcond = su.aabb_mask_and_condition(depth, o, d, aabb)
    assert not bool(mask.any()) and float(cond.abs().max()) == 0
    assert torch.equal(mask.cpu(), rmask) and torch.equal(cond.cpu(), rcond)


def test_render_camera_three_tuple(gpu):
    """render_camera end to end (render -> mask -> condition) on the synthetic scene, default generator config."""
    cfg = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=32)
    model, sd = make_model(cfg, gpu)
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], 130.0, 130.0, 48.0, 48.0, 96, 96).to(gpu)
    gen = DatasetGeneratorConfig(aabb_min=[-0.25, -0.25, -0.25], aabb_max=[0.25, 0.25, 0.25], mask_dialation=(11, 11))
    rgb, mask, cond = render_camera(gen, model, cams[0])
    assert model.training                                           # graph.eval() ... graph.train() as in the reference
    bundle = cams[0].generate_rays(0)
    depth = model.eval().get_outputs_for_camera_ray_bundle(bundle)["depth"]
    rmask, rcond = su.aabb_mask_and_condition(depth.cpu(), bundle.origins.cpu(), bundle.directions.cpu(),
                                              torch.tensor([gen.aabb_min, gen.aabb_max]), gen.mask_dialation)
    assert rgb.shape == (96, 96, 3) and torch.equal(mask.cpu(), rmask) and torch.equal(cond.cpu(), rcond)
    assert len(render_camera(gen, model, cams[0], with_mask=False)) == 4   # the reference's early-exit arity


def test_config5_generator_loop_reduced(gpu):
    """BASELINE.json configs[4] at reduced size: 8 reference cameras (circle_poses) + 50 random_sphere_poses views, each
    rendered + masked + conditioned; checked view by view against the single-camera path and, for three views, against the
    oracle end to end (render RMSE gate + exact mask given the GPU depth)."""
    from signerf_amd import random_sphere_poses, sheet

    cfg = small_config()  # proposal path, small tables
    model, sd = make_model(cfg, gpu, density_bias=5.0)  # denser scene: median depth ~0.45, i.e. inside the box below
    torch.manual_seed(1)
    c2w = torch.cat([scene.benchmark_cameras(8), random_sphere_poses(50, torch.device("cpu"), 0.5, (30.0, 120.0), (0.0, 360.0),
                                                                   [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])])
    H, W = 40, 56
    cams = Cameras(c2w[:, :3], 75.0, 75.0, W / 2, H / 2, W, H).to(gpu)
    gen = DatasetGeneratorConfig(aabb_min=[-0.2, -0.2, -0.2], aabb_max=[0.2, 0.2, 0.2], mask_dialation=(7, 7))
    tiles = sheet.render_views(model, cams, gen)
    assert tiles.shape == (58, H, W, 5) and torch.isfinite(tiles).all()
    for i in (0, 7, 8, 33, 57):
        rgb, mask, cond = render_camera(gen, model, cams[i])
        assert torch.equal(tiles[i, ..., :3], rgb) and torch.equal(tiles[i, ..., 3:4], mask.float()) and torch.equal(tiles[i, ..., 4:5], cond)
    from helpers import oracle_config, rmse
    for i in (3, 20, 41):
        b = cams[i].generate_rays(0)
        ref = onf.get_outputs_for_camera_ray_bundle(sd, oracle_config(cfg), b.origins.cpu(), b.directions.cpu())
        assert rmse(tiles[i, ..., :3], ref["rgb"]) <= 1e-3
    assert 0.0 < float(tiles[..., 3].mean()) < 1.0
