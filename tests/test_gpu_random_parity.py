"""Seeded random scenarios of the render path against the CPU oracle: frame shapes down to 1x1, sample counts down to 1, cameras anywhere
around (and inside) the scene with arbitrary orientation, axis-parallel rays, random near / far planes, random render boxes (per-ray
nears / fars, rays that miss carry the 1e10 sentinel), with and without proposal nets.  The hand-picked cases of test_gpu_render.py cover the
BASELINE configurations; this sweep is there for the shapes nobody picked.  Gate: RMSE <= 1e-3 on rgb / median depth / accumulation over the
pixels whose reference is finite, identical NaN pattern elsewhere; median-depth flips (a 0.5 crossing decided differently moves the depth
by a whole bin) are counted and bounded."""
import math

import pytest
import torch

from helpers import make_model, oracle_config, small_config
from oracle import nerfacto as onf
from signerf_amd import Cameras, SceneBox

pytestmark = pytest.mark.gpu


def _random_c2w(g):
    q = torch.randn(4, generator=g)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    t = (torch.rand(3, generator=g) - 0.5) * 3.0
    return torch.cat([R, t[:, None]], dim=1)


def _look_at(pos, target):
    """c2w [3,4] of a camera at `pos` looking at `target` (nerfstudio convention: the camera looks along -z)."""
    z = pos - target
    z = z / z.norm()
    up = torch.tensor([0.0, 0.0, 1.0]) if abs(float(z[2])) < 0.95 else torch.tensor([0.0, 1.0, 0.0])
    x = torch.linalg.cross(up, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    return torch.stack([x, y, z, pos], dim=1)


def _compare(model, sd, cfg, bundle, tag, min_finite_depth=0.0):
    out = model.get_outputs_for_camera_ray_bundle(bundle)
    n = None if bundle.nears is None else bundle.nears.cpu()
    f = None if bundle.fars is None else bundle.fars.cpu()
    ref = onf.get_outputs_for_camera_ray_bundle(sd, oracle_config(cfg), bundle.origins.cpu(), bundle.directions.cpu(), n, f)
    msgs = []
    for k in ("rgb", "depth", "accumulation"):
        got, want = out[k].cpu(), ref[k]
        assert got.shape == want.shape, (tag, k)
        ok = torch.isfinite(want)
        assert torch.equal(torch.isfinite(got), ok), f"{tag}: {k}: the non-finite pixels differ"
        if k == "depth":
            assert float(ok.float().mean()) >= min_finite_depth, f"{tag}: only {float(ok.float().mean()):.2f} of the rays hit the render box: vacuous"
        if not bool(ok.any()):
            continue
        d = (got[ok].double() - want[ok].double())
        if k == "depth":
            flips = (d.abs() / want[ok].double().abs().clamp_min(1e-6)) > 1e-3     # a whole-bin jump
            assert int(flips.sum()) <= max(1, ok.sum().item() // 500), f"{tag}: {int(flips.sum())} median-depth flips of {int(ok.sum())}"
            d = d[~flips]
            # depth lives on [0, far]: the absolute gate for depths O(1), the relative one beyond
            w = want[ok].double()[~flips]
            err = float(torch.sqrt(torch.mean((d / w.abs().clamp_min(1.0)) ** 2))) if d.numel() else 0.0
        else:
            err = float(torch.sqrt(torch.mean(d ** 2)))
        msgs.append(f"{k} {err:.1e}")
        assert err <= 1e-3, f"{tag}: {k} rmse {err:.2e}"
    return ", ".join(msgs)


@pytest.mark.parametrize("seed", range(10))
def test_random_uniform_sampler_scenarios(gpu, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    S = [1, 2, 3, 7, 16, 33, 40, 64, 5, 24][seed]
    cfg = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=S, far_plane=[1000.0, 1000.0, 50.0, 7.0, 1000.0, 300.0, 1000.0, 2.5, 1000.0, 20.0][seed])
    model, sd = make_model(cfg, gpu, seed=seed, density_bias=[4.0, 5.0, 3.0, 6.0, 4.0, 2.0, 5.0, 7.0, 4.0, 4.5][seed])
    for trial in range(3):
        H, W = int(torch.randint(1, 71, (1,), generator=g)), int(torch.randint(1, 71, (1,), generator=g))
        focal = float(torch.rand(1, generator=g) * 80 + 10)
        c2w = _random_c2w(g)
        if trial == 2:   # axis-aligned camera: direction components that are exactly 0 (slab tests divide by d + 1e-6, contraction by |p|)
            c2w = torch.tensor([[1.0, 0, 0, 0.3], [0, 1.0, 0, -0.2], [0, 0, 1.0, 1.1]])
        cams = Cameras(c2w[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        box = None
        if trial == 1:   # a random render box, the camera aimed at it from a random position outside: some rays hit it, some miss (sentinel)
            lo = (torch.rand(3, generator=g) - 1.0) * 0.4
            hi = lo + torch.rand(3, generator=g) * 0.6 + 0.05
            box = SceneBox(aabb=torch.stack([lo, hi]))
            pos = (lo + hi) / 2 + torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * float(torch.rand(1, generator=g) * 1.0 + 0.8)
            cams = Cameras(_look_at(pos, (lo + hi) / 2)[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        model.render_aabb = box
        bundle = cams[0].generate_rays(camera_indices=0, aabb_box=box)
        model.eval()
        print(f"seed {seed} trial {trial}: {H}x{W}x{S}, far {cfg.far_plane}, box {box is not None}: " +
              _compare(model, sd, cfg, bundle, f"seed {seed} trial {trial}", min_finite_depth=0.02 if box is not None else 0.0))
    model.render_aabb = None


@pytest.mark.parametrize("seed", range(6))
def test_random_proposal_scenarios(gpu, seed):
    g = torch.Generator().manual_seed(2000 + seed)
    n0, n1, S = [(256, 96, 48), (2, 2, 1), (17, 9, 5), (64, 32, 24), (128, 3, 40), (5, 60, 8)][seed]
    iters = 1 if seed == 4 else 2
    cfg = small_config(num_proposal_iterations=iters, num_proposal_samples_per_ray=(n0, n1)[:iters] if iters == 2 else (n0,), num_nerf_samples_per_ray=S)
    model, sd = make_model(cfg, gpu, seed=seed)
    for trial in range(2):
        H, W = int(torch.randint(1, 49, (1,), generator=g)), int(torch.randint(1, 49, (1,), generator=g))
        focal = float(torch.rand(1, generator=g) * 60 + 15)
        cams = Cameras(_random_c2w(g)[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        box = None
        if trial == 1 and seed % 2 == 0:
            box = SceneBox(aabb=torch.tensor([[-0.3, -0.25, -0.2], [0.2, 0.3, 0.25]]))
            pos = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * 1.2
            cams = Cameras(_look_at(pos, torch.zeros(3))[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        model.render_aabb = box
        bundle = cams[0].generate_rays(camera_indices=0, aabb_box=box)
        model.eval()
        print(f"seed {seed} trial {trial}: {H}x{W}, samples {(n0, n1)[:iters]} + {S}, box {box is not None}: " +
              _compare(model, sd, cfg, bundle, f"seed {seed} trial {trial}", min_finite_depth=0.02 if box is not None else 0.0))
    model.render_aabb = None


@pytest.mark.parametrize("seed", range(4))
def test_random_normals_scenarios(gpu, seed):
    """The `predict_normals` outputs (kernel K3) in random scenarios, with and without a render box whose rays partly miss (NaN positions),
    uniform sampler (identical bins: the plain gate applies to both outputs)."""
    import dataclasses

    from helpers import rmse

    g = torch.Generator().manual_seed(5000 + seed)
    S = [24, 7, 40, 16][seed]
    cfg = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=S, far_plane=[1000.0, 6.0, 1000.0, 30.0][seed])
    model, sd = make_model(cfg, gpu, seed=seed)
    ocfg = dataclasses.replace(oracle_config(cfg), predict_normals=True)
    for trial in range(2):
        H, W = int(torch.randint(2, 50, (1,), generator=g)), int(torch.randint(2, 50, (1,), generator=g))
        focal = float(torch.rand(1, generator=g) * 60 + 20)
        box = None
        if trial == 1:
            box = SceneBox(aabb=torch.tensor([[-0.3, -0.25, -0.2], [0.2, 0.3, 0.25]]))
            pos = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * 1.2
            cams = Cameras(_look_at(pos, torch.zeros(3))[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        else:
            pos = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * 0.6
            cams = Cameras(_look_at(pos, torch.zeros(3))[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        model.render_aabb = box
        b = cams[0].generate_rays(camera_indices=0, aabb_box=box)
        model.eval()
        out = model.get_outputs_for_camera_ray_bundle(b)
        n_ = None if b.nears is None else b.nears.cpu()
        f_ = None if b.fars is None else b.fars.cpu()
        ref = onf.get_outputs_for_camera_ray_bundle(sd, ocfg, b.origins.cpu(), b.directions.cpu(), n_, f_)
        hit = torch.ones(H, W, dtype=torch.bool) if box is None else (f_ < 1e9).squeeze(-1)
        assert int(hit.sum()) >= 4
        msg = []
        for k in ("normals", "pred_normals"):
            got, want = out[k].cpu(), ref[k]
            e = rmse(got[hit], want[hit])
            msg.append(f"{k} {e:.1e}")
            assert e <= 1e-3, f"seed {seed} trial {trial}: {k} rmse {e:.2e} on the hit rays"
            assert torch.equal(torch.isnan(got[~hit]), torch.isnan(want[~hit])), f"seed {seed} trial {trial}: {k}: NaN pattern of the missing rays differs"
        print(f"seed {seed} trial {trial}: {H}x{W}x{S}, box {box is not None} ({int(hit.sum())} hits): " + ", ".join(msg))
    model.render_aabb = None


@pytest.mark.parametrize("seed", range(4))
def test_random_tcnn_scenarios(gpu, seed):
    """tiny-cuda-nn grid semantics (`implementation="tcnn"`) in random scenarios incl. a render box, proposal path on two of the four seeds."""
    from helpers import oracle_params_from_tcnn, synthetic_tcnn_checkpoint

    g = torch.Generator().manual_seed(6000 + seed)
    props = seed % 2 == 1
    kw = dict(num_proposal_samples_per_ray=(40, 20), num_nerf_samples_per_ray=12) if props else dict(num_proposal_iterations=0, num_nerf_samples_per_ray=[20, 0, 33, 0][seed])
    cfg = small_config(implementation="tcnn", average_init_density=3.0, **kw)
    sd = synthetic_tcnn_checkpoint(cfg, seed=seed)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    model = model.to(gpu).eval()
    params = oracle_params_from_tcnn(sd, cfg)
    for trial in range(2):
        H, W = int(torch.randint(1, 49, (1,), generator=g)), int(torch.randint(1, 49, (1,), generator=g))
        focal = float(torch.rand(1, generator=g) * 60 + 15)
        box = SceneBox(aabb=torch.tensor([[-0.3, -0.25, -0.2], [0.2, 0.3, 0.25]])) if trial == 1 else None
        pos = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * (1.2 if box is not None else 0.7)
        cams = Cameras(_look_at(pos, torch.zeros(3))[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        model.render_aabb = box
        bundle = cams[0].generate_rays(camera_indices=0, aabb_box=box)
        print(f"tcnn seed {seed} trial {trial}: {H}x{W}, proposals {props}, box {box is not None}: " +
              _compare(model, params, cfg, bundle, f"tcnn seed {seed} trial {trial}", min_finite_depth=0.02 if box is not None else 0.0))
    model.render_aabb = None


@pytest.mark.parametrize("seed", range(3))
def test_random_viewer_crop_with_proposals(gpu, seed):
    """`Model.get_outputs_for_camera(camera, obb_box)` (the viewer's crop box) with proposal nets: rays that miss the oriented box are NaN
    lanes inside the proposal kernel's waves (the r03 regression), at the viewer's small frame sizes (where K1's split-depth tail is active)."""
    import math

    from signerf_amd import OrientedBox

    g = torch.Generator().manual_seed(7000 + seed)
    cfg = small_config(num_proposal_samples_per_ray=[(64, 32), (48, 24), (96, 40)][seed], num_nerf_samples_per_ray=[24, 16, 20][seed], predict_normals=False)
    model, sd = make_model(cfg, gpu, seed=seed)
    a = float(torch.rand(1, generator=g) * math.pi)
    R = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
    box = OrientedBox(R=R, T=(torch.rand(3, generator=g) - 0.5) * 0.1, S=torch.rand(3, generator=g) * 0.3 + 0.15)
    H, W = int(torch.randint(24, 80, (1,), generator=g)), int(torch.randint(24, 80, (1,), generator=g))
    pos = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * 1.0
    cam = Cameras(_look_at(pos, box.T)[None], 1.1 * W, 1.1 * W, W / 2, H / 2, W, H).to(gpu)[0]
    b = cam.generate_rays(0, obb_box=box)
    model.eval()
    out = model.get_outputs_for_camera(cam, obb_box=box)
    ref = onf.get_outputs_for_camera_ray_bundle(sd, oracle_config(cfg), b.origins.cpu(), b.directions.cpu(), b.nears.cpu(), b.fars.cpu())
    hit = (b.fars.cpu() < 1e9)
    frac = float(hit.float().mean())
    assert 0.03 < frac < 0.98, frac
    for k, c in (("rgb", 3), ("depth", 1), ("accumulation", 1)):
        h = hit.expand(-1, -1, c)
        d = (out[k].cpu()[h].double() - ref[k][h].double())
        if k == "depth":
            d = d[(d.abs() / ref[k][h].double().abs().clamp_min(1e-6)) <= 1e-3]     # (median flips are whole-bin jumps: counted elsewhere)
        e = float(torch.sqrt(torch.mean(d ** 2)))
        assert e <= 1e-3, f"seed {seed}: {k} rmse {e:.2e} on the {int(hit.sum())} rays inside the crop"
    print(f"viewer crop seed {seed}: {H}x{W}, {frac:.2f} of the rays inside the box: ok")


@pytest.mark.parametrize("seed", range(8))
def test_random_option_scenarios(gpu, seed):
    """The r03 options in random scenarios: every combination of {piecewise, uniform} initial sampler x {contraction, scene-box
    normalisation} x {last_sample, white} background, with and without proposal nets, random frame shapes / cameras / far planes / scene
    boxes, and a render box in the second trial (per-ray nears / fars, rays that miss)."""
    g = torch.Generator().manual_seed(7000 + seed)
    sampler = "uniform" if seed & 1 else "piecewise"
    no_contract = bool(seed & 2)
    background = "white" if seed & 4 else "last_sample"
    props = seed in (1, 2, 5, 7)
    far = [1000.0, 5.0, 40.0, 6.0, 12.0, 4.0, 1000.0, 8.0][seed]
    kw = dict(far_plane=far, proposal_initial_sampler=sampler, disable_scene_contraction=no_contract, background_color=background)
    cfg = small_config(num_proposal_samples_per_ray=[(64, 32), (40, 17)][seed % 2], num_nerf_samples_per_ray=[24, 11][seed % 2], **kw) if props else \
        small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=[33, 48, 20, 64][seed % 4], **kw)
    lo = -1.0 - torch.rand(3, generator=g) * 0.5
    hi = 1.0 + torch.rand(3, generator=g) * 0.5
    scene_box = SceneBox(aabb=torch.stack([lo, hi]))
    from signerf_amd import scene as _scene

    sd = _scene.synthetic_state_dict(cfg, seed=seed, density_bias=2.0 if (sampler == "uniform" or no_contract) else 4.0)
    model = cfg.setup(scene_box=scene_box)
    model.load_state_dict(sd, strict=False)
    model.field.embedding_appearance.embedding.weight.data.copy_(sd["field.embedding_appearance.embedding.weight"])
    model = model.to(gpu).eval()
    ocfg = oracle_config(cfg, scene_aabb=scene_box.aabb.tolist())
    for trial in range(2):
        H, W = int(torch.randint(1, 57, (1,), generator=g)), int(torch.randint(1, 57, (1,), generator=g))
        focal = float(torch.rand(1, generator=g) * 60 + 15)
        pos = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * float(torch.rand(1, generator=g) * 0.8 + 0.6)
        cams = Cameras(_look_at(pos, (torch.rand(3, generator=g) - 0.5) * 0.4)[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        box = None
        if trial == 1:
            blo = (torch.rand(3, generator=g) - 1.0) * 0.4
            box = SceneBox(aabb=torch.stack([blo, blo + torch.rand(3, generator=g) * 0.6 + 0.1]))
            cams = Cameras(_look_at(pos, box.aabb.mean(0))[None], focal, focal, W / 2, H / 2, W, H).to(gpu)
        model.render_aabb = box
        bundle = cams[0].generate_rays(camera_indices=0, aabb_box=box)
        out = model.get_outputs_for_camera_ray_bundle(bundle)
        n = None if bundle.nears is None else bundle.nears.cpu()
        f = None if bundle.fars is None else bundle.fars.cpu()
        ref = onf.get_outputs_for_camera_ray_bundle(sd, ocfg, bundle.origins.cpu(), bundle.directions.cpu(), n, f)
        msgs = []
        for k in ("rgb", "depth", "accumulation"):
            got, want = out[k].cpu(), ref[k]
            ok = torch.isfinite(want)
            assert torch.equal(torch.isfinite(got), ok), f"seed {seed} trial {trial}: {k}: the non-finite pixels differ"
            if not bool(ok.any()):
                continue
            d = got[ok].double() - want[ok].double()
            if k == "depth":
                flips = (d.abs() / want[ok].double().abs().clamp_min(1e-6)) > 1e-3
                assert int(flips.sum()) <= max(1, ok.sum().item() // 300), f"seed {seed} trial {trial}: {int(flips.sum())} median-depth flips of {int(ok.sum())}"
                d = (d / want[ok].double().abs().clamp_min(1.0))[~flips]
            err = float(torch.sqrt(torch.mean(d ** 2))) if d.numel() else 0.0
            msgs.append(f"{k} {err:.1e}")
            assert err <= 1e-3, f"seed {seed} trial {trial}: {k} rmse {err:.2e}"
        print(f"seed {seed} trial {trial}: {sampler}, box-normalised {no_contract}, {background}, proposals {props}, {H}x{W}, far {far}, render box {box is not None}: " + ", ".join(msgs))
    model.render_aabb = None
