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
