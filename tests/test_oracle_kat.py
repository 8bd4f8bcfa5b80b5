"""Analytic known-answer tests for the nerfacto oracle (SURVEY.md §8(c) last row).  The reference ships no tests for
this path, so these are what anchors ``oracle.nerfacto`` (its fidelity to nerfstudio 1.0.2 itself stays UNPINNED)."""
import math

import numpy as np
import torch

from helpers import oracle_config, small_config
from oracle import nerfacto as onf
from oracle import signerf_utils as su
from signerf_amd import scene


def test_spacing_functions():
    x = torch.tensor([0.0, 1.0, 1000.0, 0.5, 2.0])
    s = onf.spacing_fn(x)
    assert torch.allclose(s, torch.tensor([0.0, 0.5, 0.9995, 0.25, 0.75]), atol=1e-7)
    assert torch.allclose(onf.spacing_fn_inv(s), x, rtol=1e-4)


def test_initial_sampler_bins():
    nears, fars = torch.zeros(3, 1), torch.full((3, 1), 1000.0)
    sb, eb = onf.initial_sampler(nears, fars, 64)
    assert sb.shape == (1, 65) and eb.shape == (3, 65)
    assert eb[0, 0] == 0 and abs(float(eb[0, -1]) - 1000.0) < 0.5
    assert torch.all(eb[:, 1:] > eb[:, :-1])
    # first half of s-space is linear in distance: bin k -> 2 * k/64 * 0.9995
    assert abs(float(eb[0, 16]) - 2 * 16 / 64 * 0.9995) < 1e-6


def test_contraction():
    inside = torch.tensor([[0.3, -0.9, 0.1]])
    assert torch.equal(onf.contract_inf(inside), inside)
    far = torch.tensor([[1e6, 2.0, -3.0]])
    assert abs(float(onf.contract_inf(far).abs().max()) - 2.0) < 1e-5
    q, sel = onf.normalized_positions(torch.tensor([[0.0, 0.0, 0.0], [1e9, 0.0, 0.0]]))
    assert torch.allclose(q[0], torch.tensor([0.5, 0.5, 0.5])) and bool(sel[0])
    assert q[1, 0] <= 1.0


def test_hash_scalings_and_hand_hash():
    sc = onf.hash_scalings(16, 16, 2048)
    assert sc[0] == 16 and sc.shape == (16,) and 2040 <= float(sc[-1]) <= 2048
    assert torch.all(sc[1:] > sc[:-1])
    # hand-computed table rows for integer corners, T = 2**19
    T = 2**19
    for (x, y, z) in [(0, 0, 0), (1, 0, 0), (3, 5, 7), (2047, 2047, 2047)]:
        want = ((x * 1) ^ (y * 2654435761) ^ (z * 805459861)) % T
        got = onf.hash_fn(torch.tensor([[x, y, z]], dtype=torch.int32)[None], T, torch.zeros(1, dtype=torch.int64))
        assert int(got) == want
    # a point on a grid vertex: ceil == floor, offset 0, value = that vertex's entry
    table = torch.arange(2 * T * 2, dtype=torch.float32).view(2 * T, 2)
    sc2 = torch.tensor([16.0, 32.0])
    q = torch.tensor([[0.25, 0.5, 0.75]])
    enc = onf.hash_encode(q, table, sc2, 19)
    for lvl, s in enumerate((16, 32)):
        x, y, z = int(0.25 * s), int(0.5 * s), int(0.75 * s)
        row = ((x * 1) ^ (y * 2654435761) ^ (z * 805459861)) % T + lvl * T
        assert torch.equal(enc[0, 2 * lvl : 2 * lvl + 2], table[row])


def test_hash_trilinear_weights_sum_to_one():
    T = 2**10
    table = torch.ones(3 * T, 2)
    enc = onf.hash_encode(torch.rand(100, 3), table, torch.tensor([16.0, 21.0, 30.0]), 10)
    assert torch.allclose(enc, torch.ones_like(enc), atol=1e-6)


def test_sh_closed_form():
    d = torch.tensor([[1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0], [0, -1.0, 0], [0, 0, 1.0], [0, 0, -1.0]])
    c = onf.sh_components(d, 4)
    assert torch.allclose(c[:, 0], torch.full((6,), 0.28209479))
    assert abs(float(c[0, 3]) - 0.48860251) < 1e-7 and abs(float(c[1, 3]) + 0.48860251) < 1e-7
    assert abs(float(c[2, 1]) - 0.48860251) < 1e-7 and abs(float(c[4, 2]) - 0.48860251) < 1e-7
    assert abs(float(c[4, 6]) - (0.94617470 - 0.31539157)) < 1e-6
    assert abs(float(c[0, 8]) - 0.54627422) < 1e-7 and abs(float(c[2, 8]) + 0.54627422) < 1e-7
    assert abs(float(c[4, 12]) - 0.3731763325901154 * 2) < 1e-6
    assert abs(float(c[0, 15]) - 0.5900435899266435) < 1e-7


def test_homogeneous_medium():
    sigma, far, N = 0.7, 8.0, 4096
    bins = torch.linspace(0, far, N + 1)[None]
    starts, ends = bins[:, :-1, None], bins[:, 1:, None]
    dens = torch.full((1, N, 1), sigma)
    w = onf.get_weights(ends - starts, dens)
    acc = float(onf.render_accumulation(w))
    assert abs(acc - (1 - math.exp(-sigma * far))) < 1e-4
    depth, idx = onf.render_depth_median(w, starts, ends)
    assert abs(float(depth) - math.log(2) / sigma) < far / N
    # transmittance at sample i is exp(-sigma * t_i)
    T = w[0, :, 0] / (1 - torch.exp(-(ends - starts)[0, :, 0] * sigma))
    assert torch.allclose(T, torch.exp(-sigma * starts[0, :, 0]), atol=1e-4)


def test_empty_space_and_slab():
    N = 32
    bins = torch.linspace(0, 4, N + 1)[None]
    starts, ends = bins[:, :-1, None], bins[:, 1:, None]
    rgb = torch.rand(1, N, 3)
    w = onf.get_weights(ends - starts, torch.zeros(1, N, 1))
    assert float(w.abs().max()) == 0
    assert torch.allclose(onf.render_rgb(rgb, w), rgb[:, -1])            # background = last sample
    d, idx = onf.render_depth_median(w, starts, ends)
    assert int(idx) == N - 1                                            # searchsorted falls off the end -> clamped
    dens = torch.zeros(1, N, 1)
    dens[0, 10] = 1e4                                                   # opaque slab
    w = onf.get_weights(ends - starts, dens)
    d, idx = onf.render_depth_median(w, starts, ends)
    assert int(idx) == 10 and abs(float(onf.render_accumulation(w)) - 1) < 1e-6
    assert torch.allclose(onf.render_rgb(rgb, w), rgb[:, 10], atol=1e-6)


def test_pdf_sampler_one_hot():
    N, M = 16, 8
    sb = torch.linspace(0, 1, N + 1)[None]
    w = torch.zeros(1, N)
    w[0, 5] = 1.0
    nb, inds, cdf = onf.pdf_sample(sb, w, M, 0.01)
    assert nb.shape == (1, M + 1) and torch.all(nb[:, 1:] >= nb[:, :-1])
    # mass of interval 5 after padding = 1.01 / 1.16; every u inside that mass maps into [5/16, 6/16]
    lo, hi = 5 * 0.01 / 1.16, (5 * 0.01 + 1.01) / 1.16
    u = onf.pdf_u(M)
    inside = (u > lo) & (u < hi)
    assert int(inside.sum()) >= M - 1
    assert torch.all(nb[0, inside] >= 5 / 16 - 1e-6) and torch.all(nb[0, inside] <= 6 / 16 + 1e-6)
    # all-zero weights: uniform resampling
    nb0, _, _ = onf.pdf_sample(sb, torch.zeros(1, N), M, 0.01)
    assert torch.allclose(nb0[0], u, atol=1e-6)


def test_config1_chunk_invariance_and_ray_order():
    """BASELINE.json configs[0]: 64x64 image, 32 samples, CPU path; chunk sizes must not change the image."""
    cfg = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=32)
    ocfg = oracle_config(cfg)
    params = scene.synthetic_state_dict(cfg, seed=0)
    c2w = scene.benchmark_cameras(8)[0]
    rays = onf.generate_rays(c2w[:3], 64.0, 64.0, 32.0, 32.0, 64, 64)
    assert rays["coords"][5, 7].tolist() == [5, 7] and int(rays["camera_indices"].max()) == 0
    a = onf.get_outputs_for_camera_ray_bundle(params, ocfg, rays["origins"], rays["directions"], chunk=1024)
    b = onf.get_outputs_for_camera_ray_bundle(params, ocfg, rays["origins"], rays["directions"], chunk=4096)
    for k in ("rgb", "depth", "accumulation"):
        assert a[k].shape[:2] == (64, 64)
        assert torch.equal(a[k], b[k]), k
    # expected depth clips against chunk-global bounds, the one documented chunk-size dependence (A17)
    assert torch.allclose(a["expected_depth"], b["expected_depth"], atol=1e-5)
    # non-vacuous image (SURVEY.md §8(d) gate; accumulation saturates by construction, see signerf_amd/scene.py)
    assert float(a["rgb"].std()) > 0.05 and float(a["depth"].std()) > 0.02


def test_proposal_path_runs_and_is_monotone():
    cfg = small_config()
    ocfg = oracle_config(cfg)
    params = scene.synthetic_state_dict(cfg, seed=0)
    c2w = scene.benchmark_cameras(8)[1]
    rays = onf.generate_rays(c2w[:3], 20.0, 20.0, 8.0, 6.0, 12, 16)
    out = onf.get_outputs(params, ocfg, rays["origins"].reshape(-1, 3), rays["directions"].reshape(-1, 3), return_debug=True)
    eb = out["_debug"]["euclid_bins"]
    assert eb.shape == (192, 49) and torch.all(eb[:, 1:] >= eb[:, :-1])
    for k in ("rgb", "depth", "accumulation", "expected_depth", "prop_depth_0", "prop_depth_1"):
        assert torch.isfinite(out[k]).all(), k


def test_ellipse_element_matches_opencv_documented_shapes():
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, ...) as printed in OpenCV's morphology tutorial."""
    from oracle import signerf_utils as su

    assert su.ellipse_element(3, 3).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
    assert su.ellipse_element(5, 5).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    e = su.ellipse_element(50, 50)
    assert e.shape == (50, 50) and e[0].sum() == 1 and e[25].sum() == 50 and e[:, 25].sum() == 50
    assert (e == e[:, ::-1][:, np.r_[49, 0:49]]).mean() > 0.9  # roughly symmetric about column 25


def test_dilate_single_pixel_reproduces_the_flipped_element():
    from oracle import signerf_utils as su

    src = np.zeros((21, 23))
    src[10, 11] = 1.0
    elem = su.ellipse_element(5, 7)
    out = su.dilate(src, elem)
    # dst(y,x) = max src(y + i - ay, x + j - ax): a single source pixel paints the element reflected about its anchor
    painted = out[10 - 3 : 10 + 4, 11 - 2 : 11 + 3] > 0
    assert np.array_equal(painted, elem[::-1, ::-1].astype(bool))
    assert (out > 0).sum() == elem.sum()
    # border: a pixel in the corner only paints what falls inside
    src2 = np.zeros((6, 6))
    src2[0, 0] = 1
    assert (su.dilate(src2, su.ellipse_element(3, 3)) > 0).sum() == 3


def test_sheet_geometry_and_downscale_known_answers():
    """datasetgenerator.py:497-502, 521-534: cell windows and the pad to a multiple of 8; a factor-2 bilinear downscale with
    align_corners=False is the mean of each 2x2 block."""
    from signerf_amd.datasetgenerator import DatasetGeneratorConfig, cell_window, sheet_geometry

    cfg = DatasetGeneratorConfig(rows=2, cols=3, border_width_between_images=0)
    assert sheet_geometry(cfg, 400, 400) == (800, 1200)
    assert cell_window(cfg, 4, 400, 400) == (400, 800, 400, 800)
    cfg = DatasetGeneratorConfig(rows=2, cols=3, border_width_between_images=5)
    assert sheet_geometry(cfg, 401, 300) == (608, 1216)  # 605 -> 608, 1213 -> 1216
    assert cell_window(cfg, 5, 401, 300) == (305, 605, 812, 1213)
    t = torch.arange(4 * 6 * 1, dtype=torch.float32).reshape(4, 6, 1)
    d = su.interpolate_hwc(t, 2, 3)
    assert torch.equal(d[..., 0], torch.tensor([[3.5, 5.5, 7.5], [15.5, 17.5, 19.5]]))
    img, msk, cnd = su.compose_reference_sheet([(torch.zeros(4, 4, 3), torch.ones(4, 4, 1, dtype=torch.bool), torch.full((4, 4, 1), 0.5))], 1, 2, 2, 2, 1)
    assert img.shape == (8, 8, 3) and float(img[:2, :2].max()) == 0.0 and float(img[2:].min()) == 1.0
    assert float(msk[:2, :2].min()) == 1.0 and float(msk.sum()) == 4.0 and float(cnd[:2, :2].min()) == 0.5


def test_normals_outputs_are_unit_vectors_and_follow_finite_differences():
    """Row a16: the analytic normal is minus the normalised gradient of the pre-activation density w.r.t. the NORMALISED sample
    location; checked against central differences inside one voxel of a one-level grid (exact there: the blend is trilinear and
    the MLP piecewise linear), and the rendered outputs are unit vectors mapped to [0, 1]."""
    import dataclasses

    cfg = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=8)
    sd = scene.synthetic_state_dict(cfg, seed=3)
    ocfg = dataclasses.replace(oracle_config(cfg), predict_normals=True)
    torch.manual_seed(0)
    pos = (torch.rand(1, 64, 3) - 0.5) * 1.6   # inside the unit box: q = (p + 2) / 4, no contraction
    n = onf.field_analytic_normals(sd, ocfg, pos)
    assert torch.allclose(n.norm(dim=-1), torch.ones(1, 64), atol=1e-5)

    def h0(p):
        _, h, _, _ = onf.density_field(sd, "field.mlp_base", ocfg.main, p.float(), ocfg.average_init_density)
        return h[..., 0].double()

    eps = 2e-5   # well inside a voxel of the finest level for almost every point; q = p / 4 + 1/2
    g = torch.stack([(h0(pos + eps * e) - h0(pos - eps * e)) / (2 * eps) for e in torch.eye(3)], dim=-1) * 4.0
    fd = -torch.nn.functional.normalize(g, dim=-1).float()
    cos = (fd * n).sum(-1)
    assert float(cos.median()) > 0.99, float(cos.median())   # fp32 differences are noisy; the direction must agree

    o = torch.tensor([[0.0, 0.0, 0.6]]).repeat(4, 1)
    d = torch.nn.functional.normalize(torch.tensor([[0.1, 0.05, -1.0], [0.0, 0.2, -1.0], [-0.2, 0.0, -1.0], [0.1, -0.1, -1.0]]), dim=-1)
    out = onf.get_outputs(sd, ocfg, o, d)
    for k in ("normals", "pred_normals"):
        v = out[k] * 2 - 1
        assert out[k].shape == (4, 3) and torch.allclose(v.norm(dim=-1), torch.ones(4), atol=1e-4), k


def test_sh_basis_against_scipy_on_unit_directions():
    """The 16 real-SH polynomials of `sh_components` against an independent implementation (scipy's complex spherical harmonics,
    recombined into the real basis): component (l, m) at index l^2 + l + m must equal the real SH of that degree and order up to the
    sign convention (nerfstudio's table follows the graphics convention without the Condon-Shortley phase) -- guards the 16
    coefficients and monomials against transcription errors.  (On unit vectors only: nerfstudio's torch path feeds (d + 1) / 2, on
    which the same polynomials are evaluated.)"""
    import warnings

    import numpy as np
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from scipy.special import sph_harm

    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(500, 3, generator=g, dtype=torch.float64), dim=-1)
    comp = onf.sh_components(d, 4).numpy()
    x, y, z = d[:, 0].numpy(), d[:, 1].numpy(), d[:, 2].numpy()
    theta = np.arctan2(y, x)                 # azimuth
    phi = np.arccos(np.clip(z, -1.0, 1.0))   # polar angle
    for l in range(4):
        for m in range(-l, l + 1):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")      # (scipy deprecates sph_harm in favour of sph_harm_y; either is fine here)
                Y = sph_harm(abs(m), l, theta, phi)
            real = Y.real if m == 0 else (np.sqrt(2.0) * (Y.imag if m < 0 else Y.real))
            got = comp[:, l * l + l + m]
            err = min(np.abs(got - real).max(), np.abs(got + real).max())   # equal up to the sign convention
            assert err <= 1e-12, (l, m, err)


def test_background_colours_in_empty_space_and_behind_a_thin_medium():
    """RGBRenderer [NS], eval mode: rgb = sum w c + background (1 - sum w).  Zero weights return the background itself; weights that sum to
    0.25 leave 0.75 of it; "random" is a training device and composites like black."""
    rgb_s = torch.rand(5, 7, 3)
    zero = torch.zeros(5, 7, 1)
    assert torch.equal(onf.render_rgb(rgb_s, zero, "white"), torch.ones(5, 3))
    assert torch.equal(onf.render_rgb(rgb_s, zero, "black"), torch.zeros(5, 3))
    assert torch.equal(onf.render_rgb(rgb_s, zero, "last_sample"), rgb_s[:, -1])
    w = torch.zeros(5, 7, 1)
    w[:, 2] = 0.25
    want = 0.25 * rgb_s[:, 2] + 0.75
    assert torch.allclose(onf.render_rgb(rgb_s, w, "white"), want, atol=1e-7)
    assert torch.equal(onf.render_rgb(rgb_s, w, "random"), onf.render_rgb(rgb_s, w, "black"))


def test_uniform_sampler_and_box_normalisation_kat():
    """The two r03 options of the oracle against hand-computed values: UniformSampler's spacing is the identity, so the euclidean bins
    are the linear blend of near and far; SceneBox.get_normalized_positions is (p - aabb[0]) / (aabb[1] - aabb[0]) and the (0, 1)
    selector zeroes what falls on or outside the faces."""
    x = torch.tensor([0.0, 0.25, 1.0, 7.0, 1000.0])
    assert torch.equal(onf.spacing_fn(x, "uniform"), x) and torch.equal(onf.spacing_fn_inv(x, "uniform"), x)
    sb, eb = onf.initial_sampler(torch.tensor([[1.0], [0.0]]), torch.tensor([[5.0], [8.0]]), 4, "uniform")
    assert torch.equal(sb, torch.tensor([[0.0, 0.25, 0.5, 0.75, 1.0]]))
    assert torch.equal(eb, torch.tensor([[1.0, 2.0, 3.0, 4.0, 5.0], [0.0, 2.0, 4.0, 6.0, 8.0]]))
    # the same bins through the default sampler are not linear
    _, eb_pw = onf.initial_sampler(torch.tensor([[0.0]]), torch.tensor([[8.0]]), 4)
    assert not torch.allclose(eb_pw, eb[1:])
    aabb = torch.tensor([[-1.0, -2.0, 0.0], [3.0, 2.0, 0.5]])
    p = torch.tensor([[1.0, 0.0, 0.25], [-1.0, 0.0, 0.25], [3.5, 0.0, 0.25], [0.0, -1.0, 0.125]])
    q, sel = onf.normalized_positions(p, aabb)
    assert sel.tolist() == [True, False, False, True]                    # on a face (q = 0) and beyond one (q > 1): dropped
    assert torch.equal(q, torch.tensor([[0.5, 0.5, 0.5], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.25, 0.25, 0.25]]))
    cfg = onf.NerfactoConfig(disable_scene_contraction=True, scene_aabb=((-1.0, -2.0, 0.0), (3.0, 2.0, 0.5)))
    assert torch.equal(onf.scene_aabb(cfg), aabb) and onf.scene_aabb(onf.NerfactoConfig()) is None
