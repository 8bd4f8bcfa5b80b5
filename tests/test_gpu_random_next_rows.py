"""Seeded random scenarios for the steps either side of the render: the mask + condition step against the oracle's
restatement of datasetgenerator.py:758-818 (bit-exact), the bilinear resize against torch's own F.interpolate (the call the reference makes),
the uint8 conversion against image_tensor_converter.py's truncation.  The parametrised cases next door cover the reference's defaults; this
sweep is for the shapes nobody picked: 1-pixel frames, elements wider than the image, even / odd / non-square elements, boxes behind the camera,
depth maps with NaN / inf (what a ray that misses render_aabb yields), windows with odd strides."""
import os

import pytest
import torch

from oracle import nerfacto as onf
from oracle import signerf_utils as su
from signerf_amd.dataset_io import tensor_to_uint8
from signerf_amd.datasetgenerator import aabb_mask_and_condition
from signerf_amd.ops import resize_bilinear
from test_gpu_random_parity import _look_at, _random_c2w

pytestmark = pytest.mark.gpu
EXTRA = int(os.environ.get("SN_SOAK_EXTRA", "0"))   # a soak run appends this many seeds to every sweep (the fixed lists wrap around)


@pytest.mark.parametrize("seed", range(12 + EXTRA))
def test_random_mask_and_condition(gpu, seed):
    g = torch.Generator().manual_seed(3000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    H, W = (1, 1) if seed == 0 else ((1, 37) if seed == 1 else (ri(2, 150), ri(2, 150)))
    lo = (torch.rand(3, generator=g) - 0.5) * 0.4
    aabb = torch.stack([lo, lo + torch.rand(3, generator=g) * 0.3 + 0.02])
    centre = aabb.mean(0)
    pos = centre + torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * float(torch.rand(1, generator=g) * 1.0 + 0.4)
    c2w = _look_at(pos, centre) if seed % 4 != 3 else _random_c2w(g)          # every fourth camera looks anywhere (often past the box)
    focal = float(torch.rand(1, generator=g) * 1.5 + 0.5) * max(W, 8)
    r = onf.generate_rays(c2w, focal, focal, W / 2, H / 2, H, W)
    nears, fars = su.intersect_with_aabb(r["origins"], r["directions"], aabb)
    # a depth map that puts some pixels inside the box, some in front, some behind; a few NaN / inf entries
    u = torch.rand(H, W, 1, generator=g)
    depth = torch.where(u < 0.5, nears + (fars - nears) * torch.rand(H, W, 1, generator=g), torch.rand(H, W, 1, generator=g) * 3.0)
    if H * W > 4:
        depth.view(-1)[ri(0, H * W - 1)] = float("nan")
        depth.view(-1)[ri(0, H * W - 1)] = float("inf")
    dil = [None, (1, 1), (2, 2), (3, 3), (50, 50), (7, 21), (20, 20), (64, 3), (5, 5), (151, 151), (4, 9), (11, 11)][seed % 12]
    inverse = seed % 12 in (4, 9) or seed % 7 == 6
    manual = (0.2, 1.7) if seed % 12 in (5, 10) else None
    radius = [0.1, 0.0, 0.25][seed % 3]
    rmask, rcond = su.aabb_mask_and_condition(depth, r["origins"], r["directions"], aabb, dil, inverse, manual, radius)
    mask, cond = aabb_mask_and_condition(depth.to(gpu), r["origins"].to(gpu), r["directions"].to(gpu), aabb, dil, inverse, manual, radius)
    print(f"seed {seed}: {H}x{W}, element {dil}, inverse {inverse}, manual {manual}: mask coverage {float(rmask.float().mean()):.3f}")
    assert torch.equal(mask.cpu(), rmask), f"mask differs in {int((mask.cpu() != rmask).sum())} pixels"
    a, b = torch.nan_to_num(cond.cpu(), nan=-7.0), torch.nan_to_num(rcond, nan=-7.0)
    bad = torch.nonzero((a != b).view(-1)).flatten()
    assert bad.numel() == 0, (f"condition image differs in {bad.numel()} pixels; first: got {a.view(-1)[bad[:4]].tolist()} want {b.view(-1)[bad[:4]].tolist()} "
                              f"depth {depth.view(-1)[bad[:4]].tolist()}")


@pytest.mark.parametrize("seed", range(10 + EXTRA))
def test_random_resize_windows(gpu, seed):
    g = torch.Generator().manual_seed(4000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    C = [1, 3, 3, 2, 1, 3, 4, 3, 1, 3][seed % 10]
    BH, BW = ri(4, 200), ri(4, 200)
    big = torch.rand(BH, BW, C, generator=g)
    y0, x0 = ri(0, BH - 1), ri(0, BW - 1)
    h, w = ri(1, BH - y0), ri(1, BW - x0)
    oh, ow = ri(1, 160), ri(1, 160)
    SH, SW = oh + ri(0, 20), ow + ri(0, 20)
    sy, sx = ri(0, SH - oh), ri(0, SW - ow)
    sheet = torch.full((SH, SW, C), -3.0)
    d_sheet = sheet.to(gpu)
    src = big.to(gpu)[y0:y0 + h, x0:x0 + w, :]
    resize_bilinear(src, oh, ow, out=d_sheet[sy:sy + oh, sx:sx + ow, :])
    ref = sheet.clone()
    ref[sy:sy + oh, sx:sx + ow, :] = su.interpolate_hwc(big[y0:y0 + h, x0:x0 + w, :].contiguous(), oh, ow)
    err = float((d_sheet.cpu() - ref).abs().max())
    print(f"seed {seed}: {h}x{w}x{C} window at ({y0},{x0}) of {BH}x{BW} -> {oh}x{ow} at ({sy},{sx}) of {SH}x{SW}: max err {err:.1e}")
    assert err <= 5e-7
    outside = torch.ones(SH, SW, dtype=torch.bool)
    outside[sy:sy + oh, sx:sx + ow] = False
    assert bool((d_sheet.cpu()[outside] == -3.0).all())                      # nothing outside the destination window is touched
    # the uint8 source form (the 0/1 mask) and the thresholded output
    m = (torch.rand(h, w, 1, generator=g) > 0.5)
    got = resize_bilinear(m.to(gpu), oh, ow, threshold=True).cpu()
    want = su.interpolate_hwc(m.float(), oh, ow)
    near_half = (want - 0.5).abs() < 1e-6
    assert torch.equal((got > 0.5)[~near_half], (want > 0.5)[~near_half])


def test_random_tensor_to_uint8(gpu):
    """image_tensor_converter.py:22-23: (x * 255).astype(uint8) -- truncation, no rounding, no clamp (values in [0, 1])."""
    g = torch.Generator().manual_seed(5)
    x = torch.rand(257, 131, 3, generator=g)
    x.view(-1)[:7] = torch.tensor([0.0, 1.0, 254.9 / 255, 0.5, 1 / 255, 0.999999, 127.5 / 255])
    assert torch.equal(tensor_to_uint8(x.to(gpu)).cpu(), torch.from_numpy(su.tensor_to_uint8(x)))


@pytest.mark.parametrize("seed", range(8 + EXTRA))
def test_random_cameras_generate_rays_and_box_bounds(gpu, seed):
    """Row a5 for arbitrary pin-hole cameras: any pose, fx != fy, principal point off centre (even outside the frame), frames down to 1x1;
    with a render box the bundle's nears / fars against nerfstudio's clamped slab test (oracle), the SIGNeRF helper `intersect_with_aabb`
    (row a4, no clamp, 1 / (d + 1e-6)) bit for bit on the same rays."""
    from signerf_amd import Cameras, SceneBox, intersect_with_aabb

    g = torch.Generator().manual_seed(8000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    H, W = [(1, 1), (1, 64), (64, 1), (7, 9)][seed] if seed < 4 else (ri(2, 120), ri(2, 120))
    fx, fy = float(torch.rand(1, generator=g) * 200 + 5), float(torch.rand(1, generator=g) * 200 + 5)
    cx, cy = float((torch.rand(1, generator=g) * 1.4 - 0.2) * W), float((torch.rand(1, generator=g) * 1.4 - 0.2) * H)
    c2w = _random_c2w(g)
    cam = Cameras(c2w[None], fx, fy, cx, cy, W, H).to(gpu)[0]
    lo = (torch.rand(3, generator=g) - 0.5) * 2.0
    box = SceneBox(aabb=torch.stack([lo, lo + torch.rand(3, generator=g) * 1.5 + 0.05]))
    b = cam.generate_rays(camera_indices=0, aabb_box=box)
    ref = onf.generate_rays(c2w, fx, fy, cx, cy, H, W)
    assert b.origins.shape == (H, W, 3) and torch.equal(b.origins.cpu(), ref["origins"])
    assert float((b.directions.cpu() - ref["directions"]).abs().max()) <= 3e-7
    assert float((b.metadata["directions_norm"].cpu() - ref["directions_norm"]).abs().max() / ref["directions_norm"].abs().max()) <= 1e-6
    o, d = b.origins.cpu().reshape(-1, 3), b.directions.cpu().reshape(-1, 3)
    tmin, tmax = onf.intersect_aabb_ns(o, d, box.aabb.flatten())
    miss = tmin >= 1e9
    assert torch.equal((b.nears.cpu().reshape(-1) >= 1e9), miss), "the rays that miss the box differ"
    assert torch.allclose(b.nears.cpu().reshape(-1)[~miss], tmin[~miss], rtol=2e-6, atol=1e-6) and torch.allclose(b.fars.cpu().reshape(-1)[~miss], tmax[~miss], rtol=2e-6, atol=1e-6)
    n, f = intersect_with_aabb(b.origins, b.directions, box.aabb)
    rn, rf = su.intersect_with_aabb(b.origins.cpu(), b.directions.cpu(), box.aabb)
    assert torch.equal(torch.nan_to_num(n.cpu(), nan=-7.0), torch.nan_to_num(rn, nan=-7.0)) and torch.equal(torch.nan_to_num(f.cpu(), nan=-7.0), torch.nan_to_num(rf, nan=-7.0))
    print(f"seed {seed}: {H}x{W}, fx {fx:.1f} fy {fy:.1f} c ({cx:.1f}, {cy:.1f}): {int((~miss).sum())} of {miss.numel()} rays hit the box")
