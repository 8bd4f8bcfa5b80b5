"""GPU parity of every stage of the path against the CPU oracle, through the C ABI.
Integer results (ray<->pixel map, hash-grid table rows, searchsorted / median indices) must be BIT-EXACT; floating point
within the tolerance written next to each assert (north_star: 1e-3 RMSE end to end)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, make_model, oracle_config, rmse, small_config
from oracle import nerfacto as onf
from oracle import signerf_utils as su
from signerf_amd import Cameras, SceneBox, intersect_with_aabb, ops, scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_small(gpu):
    cfg = small_config()
    model, sd = make_model(cfg, gpu)
    return cfg, model, sd


@pytest.fixture(scope="module")
def model_full(gpu):
    """Full-size tables (2^19 main, 2^17 proposal) -- BASELINE.json configs[3] architecture."""
    cfg = scene.proposal_config()
    model, sd = make_model(cfg, gpu)
    return cfg, model, sd


# ---- row a5 -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,W,cam", [(64, 64, 0), (37, 53, 3), (8, 200, 5)])
def test_generate_rays(gpu, H, W, cam):
    c2w = scene.benchmark_cameras(8)
    fx, fy, cx, cy = 1.2 * W, 1.1 * W, W / 2 + 0.25, H / 2 - 0.5
    cams = Cameras(c2w[:, :3], fx, fy, cx, cy, W, H).to(gpu)
    assert len(cams) == 8
    b = cams[cam].generate_rays(camera_indices=0)
    ref = onf.generate_rays(c2w[cam, :3], fx, fy, cx, cy, H, W)
    assert b.origins.shape == (H, W, 3) and b.pixel_area.shape == (H, W, 1)
    assert torch.equal(b.origins.cpu(), ref["origins"])                       # translation column, copied
    # the SIGNeRF call site passes camera_indices=0 on a 0-dim camera (datasetgenerator.py:691): indices are all 0
    assert int(b.camera_indices.max()) == 0 and int(b.camera_indices.min()) == 0 and b.camera_indices.dtype == torch.int64
    d = b.directions.cpu()
    assert float((d - ref["directions"]).abs().max()) <= 2e-7               # fp32 ulp-level
    assert float(((b.pixel_area.cpu() - ref["pixel_area"]).abs() / ref["pixel_area"]).max()) <= 1e-3
    assert float((b.metadata["directions_norm"].cpu() - ref["directions_norm"]).abs().max()) <= 1e-6
    # ray index <-> (y, x): pixel (y, x) must hold the direction through pixel centre (x+.5, y+.5) -- exact index map
    y, x = H // 3, (2 * W) // 3
    cam_dir = torch.tensor([(x + 0.5 - cx) / fx, -(y + 0.5 - cy) / fy, -1.0])
    w = c2w[cam, :3, :3] @ cam_dir
    assert torch.allclose(d[y, x], w / w.norm(), atol=1e-6)


def test_generate_rays_with_aabb(gpu):
    c2w = scene.benchmark_cameras(8)
    cams = Cameras(c2w[:, :3], 60.0, 60.0, 24.0, 24.0, 48, 48).to(gpu)
    box = SceneBox(aabb=torch.tensor([[-0.12, -0.1, -0.08], [0.1, 0.12, 0.09]]))
    b = cams[2].generate_rays(camera_indices=0, aabb_box=box)
    ref = onf.generate_rays(c2w[2, :3], 60.0, 60.0, 24.0, 24.0, 48, 48)
    tmin, tmax = onf.intersect_aabb_ns(b.origins.cpu().reshape(-1, 3), b.directions.cpu().reshape(-1, 3), box.aabb.flatten())
    assert torch.allclose(b.nears.cpu().reshape(-1), tmin, rtol=1e-6, atol=1e-6)
    assert torch.allclose(b.fars.cpu().reshape(-1), tmax, rtol=1e-6, atol=1e-6)
    assert float((b.nears == 1e10).float().mean()) > 0.05  # some rays miss the box -> sentinel


# ---- row a4 (golden, bit-exact) -----------------------------------------------------------------------------------
def test_intersect_with_aabb_golden(gpu):
    g = np.load(os.path.join(GOLDEN, "intersect_with_aabb.npz"))
    o, d = torch.tensor(g["origins"]).to(gpu), torch.tensor(g["directions"]).to(gpu)
    for box, n, f in (("aabb", "nears", "fars"), ("aabb2", "nears2", "fars2")):
        nears, fars = intersect_with_aabb(o, d, torch.tensor(g[box]))
        assert nears.shape == (o.shape[0], o.shape[1], 1)
        assert np.array_equal(nears.cpu().numpy(), g[n])
        assert np.array_equal(fars.cpu().numpy(), g[f])


def test_intersect_with_aabb_large_matches_oracle(gpu):
    g = torch.Generator().manual_seed(3)
    o = (torch.rand(300, 400, 3, generator=g) - 0.5) * 2
    d = torch.nn.functional.normalize(torch.randn(300, 400, 3, generator=g), dim=-1)
    d[0, :10] = torch.tensor([0.0, 0.0, 1.0])
    box = torch.tensor([[-0.1, -0.1, -0.1], [0.1, 0.1, 0.1]])
    n, f = intersect_with_aabb(o.to(gpu), d.to(gpu), box)
    rn, rf = su.intersect_with_aabb(o, d, box)
    assert torch.equal(n.cpu(), rn) and torch.equal(f.cpu(), rf)


# ---- row a13 --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", [-1, 0, 1])
def test_hash_encode_indices_bit_exact(model_full, gpu, which):
    cfg, model, sd = model_full
    ocfg = oracle_config(cfg)
    hc = ocfg.main if which < 0 else ocfg.proposals[which]
    prefix = "field.mlp_base" if which < 0 else f"proposal_networks.{which}.mlp_base"
    g = torch.Generator().manual_seed(11 + which)
    q = torch.rand(20000, 3, generator=g)
    q[:64] = torch.tensor([0.25, 0.5, 0.75])          # exactly on grid vertices at several levels
    q[64:128] = 0.0
    q[128:192] = torch.nextafter(torch.tensor(1.0), torch.tensor(0.0))
    feat, idx = ops.hash_encode(model, q.to(gpu), which, return_indices=True)
    sc = onf.hash_scalings(hc.num_levels, hc.base_res, hc.max_res)
    _, _, ridx, _ = onf.hash_corner_indices(q, sc, hc.log2_hashmap_size)
    assert torch.equal(idx.cpu().to(torch.int64), ridx)                       # BIT-EXACT table rows, all 8 corners, all levels
    ref = onf.hash_encode(q, sd[f"{prefix}.encoder.hash_table"], sc, hc.log2_hashmap_size)
    assert float((feat.cpu() - ref).abs().max()) <= 2e-6                      # table ~U(-1,1); blend differs by FMA rounding only


# ---- rows a9, a14, a15 ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["fp32", "fp16x2"])
def test_main_field_forward(model_full, gpu, precision):
    cfg, model, sd = model_full
    model.config.precision = precision
    ocfg = oracle_config(cfg)
    g = torch.Generator().manual_seed(5)
    n = 10000  # not a multiple of 64/256: exercises the ragged tail
    pos = (torch.rand(n, 3, generator=g) - 0.5) * 3.0
    pos[:100] *= 20.0                                   # far outside the unit box -> contraction branch
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    density, rgb = ops.field_forward(model, pos.to(gpu), dirs.to(gpu))
    model.config.precision = "fp32"
    rd, rh, _, _ = onf.density_field(sd, "field.mlp_base", ocfg.main, pos[:, None, :], ocfg.average_init_density)
    rrgb = onf.field_rgb(sd, ocfg, dirs, rh)[:, 0]
    rel = ((density.cpu() - rd[:, 0, 0]).abs() / rd[:, 0, 0].clamp_min(1e-6)).max()
    print(f"main field [{precision}]: max rel density err {float(rel):.2e}, max abs rgb err {float((rgb.cpu() - rrgb).abs().max()):.2e}")
    # fp32: exact-fp32 MFMA (fmaf chains).  fp16x2: operands split into fp16 hi+lo, lo.lo dropped (~2^-22 per product) --
    # the same bounds must hold, i.e. the split path is fp32-grade, not fp16-grade.
    assert float(rel) <= 1e-4                                                # density relative error (exp of an fp32 MLP)
    assert float((rgb.cpu() - rrgb).abs().max()) <= 2e-5                      # post-sigmoid colours
    assert float(rrgb.std()) > 0.05                                           # not vacuous


@pytest.mark.parametrize("which", [0, 1])
def test_proposal_field_forward(model_full, gpu, which):
    cfg, model, sd = model_full
    ocfg = oracle_config(cfg)
    g = torch.Generator().manual_seed(6 + which)
    pos = (torch.rand(5001, 3, generator=g) - 0.5) * 4.0
    density, _ = ops.field_forward(model, pos.to(gpu), None, which)
    rd, _, _, _ = onf.density_field(sd, f"proposal_networks.{which}.mlp_base", ocfg.proposals[which], pos[:, None, :], ocfg.average_init_density)
    rel = ((density.cpu() - rd[:, 0, 0]).abs() / rd[:, 0, 0].clamp_min(1e-6)).max()
    assert float(rel) <= 1e-4


# ---- rows a10, a17 ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,S", [(1000, 48), (257, 64), (3, 1), (64, 256)])
def test_composite(gpu, R, S):
    g = torch.Generator().manual_seed(R + S)
    bins = torch.cumsum(torch.rand(R, S + 1, generator=g) * 0.1, dim=-1)
    density = torch.exp(torch.randn(R, S, generator=g) * 2.0)
    density[: R // 4] = 0.0                                   # empty rays: acc 0, median clamps to S-1, rgb = last sample
    if R > 10 and S > 5:
        density[R // 4 : R // 4 + 5, 3] = 1e5                 # opaque slabs
    rgb_s = torch.rand(R, S, 3, generator=g)
    out = ops.composite(bins.to(gpu), density.to(gpu), rgb_s.to(gpu))
    starts, ends = bins[:, :-1, None], bins[:, 1:, None]
    w = onf.get_weights(ends - starts, density[..., None])
    assert float((out["weights"].cpu() - w[..., 0]).abs().max()) <= 2e-6
    d, idx = onf.render_depth_median(w, starts, ends)
    # median index: bit-exact except where the GPU's expf differs from libm's by an ulp right at the 0.5 crossing
    gi = out["median_index"].cpu().to(torch.int64)
    flips = int((gi != idx[:, 0]).sum())
    assert flips <= max(1, R // 500), f"{flips} median-index flips in {R} rays"
    same = gi == idx[:, 0]
    assert torch.equal(out["depth"].cpu()[same], d[same, 0])                   # same index -> identical mid-point (exact arithmetic)
    assert float((out["rgb"].cpu() - onf.render_rgb(rgb_s, w)).abs().max()) <= 1e-5
    assert float((out["accumulation"].cpu() - onf.render_accumulation(w)[:, 0]).abs().max()) <= 1e-5
    assert float((out["expected_depth"].cpu() - onf.render_depth_expected(w, starts, ends)[:, 0]).abs().max()) <= 1e-4
    empty = slice(0, R // 4)
    if R // 4 > 0:
        assert torch.all(gi[empty] == S - 1) and float(out["accumulation"][empty].abs().max()) == 0
        assert torch.allclose(out["rgb"].cpu()[empty], rgb_s[empty, -1], atol=1e-7)


def test_composite_median_bit_exact_given_identical_weights(gpu):
    """With transmittance-free inputs (density 0 except one sample) exp() plays no role: index must match exactly."""
    R, S = 500, 40
    g = torch.Generator().manual_seed(1)
    bins = torch.cumsum(torch.rand(R, S + 1, generator=g), dim=-1)
    hit = torch.randint(0, S, (R,), generator=g)
    density = torch.zeros(R, S)
    density[torch.arange(R), hit] = 1e6
    out = ops.composite(bins.to(gpu), density.to(gpu), torch.rand(R, S, 3, generator=g).to(gpu))
    assert torch.equal(out["median_index"].cpu().to(torch.int64), hit)


# ---- row a11 ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,N,M", [(300, 256, 96), (300, 96, 48), (5, 7, 3), (64, 64, 64)])
def test_pdf_sample(gpu, R, N, M):
    g = torch.Generator().manual_seed(N * M)
    sb = torch.sort(torch.rand(R, N + 1, generator=g), dim=-1).values
    sb[:, 0], sb[:, -1] = 0.0, 1.0
    w = torch.rand(R, N, generator=g) ** 4
    w[: R // 5] = 0.0                                          # zero-weight rays -> uniform pdf (padding branch)
    if R > 10:
        w[R // 5 : R // 5 + 3] = 0.0
        w[R // 5 : R // 5 + 3, N // 2] = 1.0                   # one-hot
    bins, inds = ops.pdf_sample(sb.to(gpu), w.to(gpu), M, 0.01)
    rb, rinds, rcdf = onf.pdf_sample(sb, w, M, 0.01)
    # searchsorted indices are integer work ON a cdf; the cdf itself is fp32 work whose last ulp depends on the order
    # of an fp32 sum.  Indices must therefore agree everywhere except at provable near-ties |cdf_knot - u| <= 2 ulp
    # (e.g. u = 0.5 against the knot 128/256 of a uniform pdf); count them, and require nothing else flips.
    u = onf.pdf_u(M)
    gap = (rcdf[:, None, :] - u[None, :, None]).abs().min(dim=-1).values      # [R, M+1] distance to the nearest knot
    flipped = inds.cpu().to(torch.int64) != rinds
    print(f"pdf_sample R={R} N={N} M={M}: {int(flipped.sum())} index flips, all at near-ties: "
          f"{bool((gap[flipped] <= 2.4e-7).all())}; near-ties present: {int((gap <= 2.4e-7).sum())}")
    assert bool((gap[flipped] <= 2.4e-7).all())
    assert int(flipped.sum()) <= int((gap <= 2.4e-7).sum())
    # a flipped tie still lands on (almost) the same bin value: the interpolant is continuous across a knot
    assert float((bins.cpu() - rb).abs().max()) <= 1e-5
    assert torch.all(bins[:, 1:] >= bins[:, :-1])


def test_composite_nan_and_inf_inputs(gpu):
    """torch.nan_to_num on the weights and on the per-sample colours (A10, A17): a NaN density zeroes its own and every later
    weight of the ray (the cumulative optical depth is NaN from there on), an infinite density makes the sample opaque, NaN
    colours count as 0."""
    R, S = 64, 24
    g = torch.Generator().manual_seed(9)
    bins = torch.cumsum(torch.rand(R, S + 1, generator=g) * 0.1, dim=-1)
    density = torch.exp(torch.randn(R, S, generator=g))
    density[:16, 5] = float("nan")
    density[16:32, 7] = float("inf")
    rgb_s = torch.rand(R, S, 3, generator=g)
    rgb_s[32:48, 3] = float("nan")
    out = ops.composite(bins.to(gpu), density.to(gpu), rgb_s.to(gpu))
    starts, ends = bins[:, :-1, None], bins[:, 1:, None]
    w = onf.get_weights(ends - starts, density[..., None])
    assert torch.isfinite(out["weights"]).all() and torch.isfinite(out["rgb"]).all()
    assert float((out["weights"].cpu() - w[..., 0]).abs().max()) <= 2e-6
    assert float(out["weights"][:16, 5:].abs().max()) == 0.0
    assert float((out["rgb"].cpu() - onf.render_rgb(rgb_s, w)).abs().max()) <= 1e-5
    assert float((out["accumulation"].cpu() - onf.render_accumulation(w)[:, 0]).abs().max()) <= 1e-5


# ---- r06: the main kernel's division-free position map == the literal strict one, on the hardware -------------------------------
def _positions_case(gpu, name, o, d, t0, t1, oracle_rows=0):
    got = ops.sample_positions(o, d, t0, t1)
    s, e = got["strict"].view(torch.int32), got["exact"].view(torch.int32)
    nan_s = torch.isnan(got["strict"]).any(dim=-1)
    nan_e = torch.isnan(got["exact"]).any(dim=-1)
    assert torch.equal(nan_s, nan_e), f"{name}: NaN samples differ"
    ok = ~nan_s
    bad = int((s[ok] != e[ok]).any(dim=-1).sum())
    fast_diff = int((got["fast"].view(torch.int32)[ok] != s[ok]).any(dim=-1).sum())
    print(f"{name}: {o.shape[0]} samples, exact != strict in {bad}; (fast != strict in {fast_diff}, max |dq| "
          f"{float((got['fast'][ok] - got['strict'][ok]).abs().max()):.1e}); {int(nan_s.sum())} NaN samples")
    assert bad == 0, f"{name}: sn_sample_q_exact differs from sn_sample_q in {bad} samples"
    if oracle_rows:   # the strict kernel itself against torch's CPU arithmetic (the oracle's own functions)
        n = min(oracle_rows, o.shape[0])
        pos = onf.sample_positions(o[:n].cpu(), d[:n].cpu(), t0[:n].cpu().view(n, 1, 1), t1[:n].cpu().view(n, 1, 1))[:, 0]   # one sample per ray
        q_ref, _ = onf.normalized_positions(pos)
        keep = ~torch.isnan(q_ref).any(dim=-1)
        assert torch.equal(got["strict"][:n].cpu()[keep].view(torch.int32), q_ref[keep].view(torch.int32)), f"{name}: strict kernel != torch CPU"
    return bad


def _unit(v):
    return v / v.norm(dim=-1, keepdim=True)


def test_exact_position_map_random(gpu):
    g = torch.Generator(device="cpu").manual_seed(0)
    n = 1 << 22
    o = ((torch.rand(n, 3, generator=g) - 0.5) * 1.4).to(gpu)
    d = _unit(torch.randn(n, 3, generator=g)).to(gpu)
    # sample distances of the lindisp sampler between 0 and 1000: s uniform, t = s^-1(s)
    s = torch.rand(n, generator=g)
    t0 = torch.where(s < 0.5, 2 * s, 1 / (2 - 2 * s).clamp_min(1e-3)).to(gpu)
    t1 = t0 * (1 + 0.05 * torch.rand(n, generator=g).to(gpu))
    _positions_case(gpu, "lindisp samples in [0, 1000]", o, d, t0, t1, oracle_rows=1 << 20)
    # inside the unit box (the contraction is the identity) and around its faces
    t0s = torch.rand(n, generator=g).to(gpu) * 1.5
    _positions_case(gpu, "around the unit box", o, d, t0s, t0s + 1e-3, oracle_rows=1 << 19)
    # rays along an axis, origins at 0, components that are exact ties of the max
    axis = torch.zeros(n, 3)
    axis[torch.arange(n), torch.randint(0, 3, (n,), generator=g)] = 1.0
    _positions_case(gpu, "axis-parallel rays", torch.zeros(n, 3, device=gpu), axis.to(gpu), t0, t1, oracle_rows=1 << 18)


def test_exact_position_map_every_significand_of_the_contraction_norm(gpu):
    """mag takes EVERY fp32 significand (2^23, at three exponents): o = 0, d = (1, a, b), start = 0, end = 2 mag, so p_x = mag exactly --
    the reciprocal's two Newton steps and the all-ones substitution are exercised on the hardware's own v_rcp_f32."""
    man = torch.arange(1 << 23, dtype=torch.int32, device=gpu)
    for e in (0, 3, 9):
        mag = (man | ((127 + e) << 23)).view(torch.float32)
        n = mag.numel()
        d = torch.stack([torch.ones(n, device=gpu), torch.full((n,), 0.37, device=gpu), torch.full((n,), -0.81, device=gpu)], dim=-1)
        o = torch.zeros(n, 3, device=gpu)
        t0 = torch.zeros(n, device=gpu)
        t1 = mag * 2.0
        _positions_case(gpu, f"every significand of |p|_inf at 2^{e}", o, d, t0, t1, oracle_rows=(1 << 20) if e == 0 else 0)
    # the all-ones significands themselves, every exponent the sampler can reach, other coordinates random
    g = torch.Generator(device="cpu").manual_seed(1)
    for e in range(0, 11):
        n = 1 << 16
        mag = torch.full((n,), float(np.float32(2.0 - 2.0 ** -23) * np.float32(2.0 ** e)), device=gpu)
        d = torch.cat([torch.ones(n, 1), torch.rand(n, 2, generator=g) * 2 - 1], dim=-1).to(gpu)
        _positions_case(gpu, f"all-ones significand at 2^{e}", torch.zeros(n, 3, device=gpu), d, torch.zeros(n, device=gpu), mag * 2.0, oracle_rows=n)


def test_exact_position_map_quotients_near_one_and_halfway(gpu):
    """Coordinates that nearly tie with the largest one (quotients just below a power of two: where q0 = RN(p y) can be 1.5 ulp off) and
    quotients constructed to sit next to a rounding boundary."""
    g = torch.Generator(device="cpu").manual_seed(2)
    n = 1 << 22
    mag = torch.exp(torch.rand(n, generator=g) * np.log(1500.0)).to(torch.float32)          # [1, 1500]
    j = torch.randint(0, 64, (n,), generator=g)
    py = (mag.view(torch.int32) - j.to(torch.int32)).view(torch.float32)                     # mag minus 0..63 ulp
    scale = 2.0 ** -torch.randint(0, 8, (n,), generator=g).to(torch.float32)
    pz = py * scale * torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0)               # ... and the same just below 2^-k
    # p = o + (d / 2) t with o = 0, t = 2: p = d exactly
    d = torch.stack([mag, py, pz], dim=-1).to(gpu)
    _positions_case(gpu, "near-tie coordinates", torch.zeros(n, 3, device=gpu), d, torch.zeros(n, device=gpu), torch.full((n,), 2.0, device=gpu),
                    oracle_rows=1 << 20)
    # quotient targets next to a midpoint: p = RN(m (qt + (1/2 +- tiny) ulp))
    qt = torch.rand(n, generator=g, dtype=torch.float64) * 0.98 + 0.01
    qt32 = qt.to(torch.float32).to(torch.float64)
    ulp = torch.tensor(np.spacing(qt32.numpy().astype(np.float32)).astype(np.float64))
    p = (mag.to(torch.float64) * (qt32 + ulp * (0.5 + (torch.rand(n, generator=g, dtype=torch.float64) - 0.5) * 2.0 ** -20))).to(torch.float32)
    d = torch.stack([mag, p, -p], dim=-1).to(gpu)
    _positions_case(gpu, "quotients next to a rounding boundary", torch.zeros(n, 3, device=gpu), d, torch.zeros(n, device=gpu),
                    torch.full((n,), 2.0, device=gpu), oracle_rows=1 << 20)


def test_exact_position_map_non_finite(gpu):
    """NaN / inf positions: both forms hand a NaN sample to the field (which coordinates are NaN may differ; the kernels restore the
    all-NaN sample from any NaN coordinate, sn_main.h)."""
    o = torch.tensor([[0.0, 0.0, 0.0], [float("nan"), 0.0, 0.0], [0.0, 0.0, 0.0], [0.1, 0.2, 0.3]], device=gpu).repeat(64, 1)
    d = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.6, 0.8, 0.0], [0.0, 0.0, 1.0]], device=gpu).repeat(64, 1)
    t0 = torch.tensor([1e10, 1.0, float("inf"), 3e38], device=gpu).repeat(64)
    got = ops.sample_positions(o, d, t0, t0)
    nan_s = torch.isnan(got["strict"]).any(dim=-1)
    assert torch.equal(nan_s, torch.isnan(got["exact"]).any(dim=-1))
    ok = ~nan_s
    assert torch.equal(got["strict"][ok], got["exact"][ok])
