"""The two nerfacto options r02's model refused (VERDICT r02 "What's missing" #5; SIGNeRF never overrides them, signerf_config.py:31-36):
``proposal_initial_sampler="uniform"`` (UniformSampler: s(x) = x through the whole proposal chain) and ``disable_scene_contraction=True``
(the fields normalise positions with the scene box, SceneBox.get_normalized_positions, instead of the L-inf contraction).  Both run in
their own kernel instantiations (SnRenderOpts.spacing_mode, SnFieldDesc.disable_scene_contraction; the production kernels' code is
untouched), against the CPU oracle like every other render test: rgb / accumulation / depths within 1e-3, normals too."""
import dataclasses

import pytest
import torch

from helpers import make_model, oracle_config, rmse, small_config
from oracle import nerfacto as onf
from signerf_amd import Cameras, SceneBox, scene
from signerf_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 1e-3
BOX = torch.tensor([[-1.3, -1.1, -0.9], [1.2, 1.4, 1.0]])     # a scene box that is neither a cube nor centred


def _model(cfg, gpu, scene_box=None, **scene_kw):
    sd = scene.synthetic_state_dict(cfg, seed=0, **scene_kw)
    model = cfg.setup(scene_box=scene_box) if scene_box is not None else cfg.setup()
    model.load_state_dict(sd, strict=False)
    model.field.embedding_appearance.embedding.weight.data.copy_(sd["field.embedding_appearance.embedding.weight"])
    return model.to(gpu).eval(), sd


def _pair(cfg, gpu, H, W, cam=0, focal=None, scene_box=None, render_aabb=None, normals=False, **scene_kw):
    model, sd = _model(cfg, gpu, scene_box, **scene_kw)
    focal = focal or float(W)
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], focal, focal, W / 2, H / 2, W, H).to(gpu)
    model.render_aabb = render_aabb
    bundle = cams[cam].generate_rays(camera_indices=0, aabb_box=render_aabb)
    out = model.get_outputs_for_camera_ray_bundle(bundle)
    ocfg = oracle_config(cfg, scene_aabb=None if scene_box is None else scene_box.aabb.tolist())
    if normals:
        ocfg = dataclasses.replace(ocfg, predict_normals=True)
    n = None if bundle.nears is None else bundle.nears.cpu()
    f = None if bundle.fars is None else bundle.fars.cpu()
    ref = onf.get_outputs_for_camera_ray_bundle(sd, ocfg, bundle.origins.cpu(), bundle.directions.cpu(), n, f)
    return model, out, ref


def _check(out, ref, keys=("rgb", "accumulation", "depth", "expected_depth")):
    for k in keys:
        e = rmse(torch.nan_to_num(out[k], nan=-1.0, posinf=1e9), torch.nan_to_num(ref[k], nan=-1.0, posinf=1e9))
        print(f"{k}: rmse {e:.2e}")
        assert out[k].shape == ref[k].shape and e <= TOL, k
    assert float(torch.nan_to_num(ref["rgb"]).std()) > 0.04 and 0.05 < float(ref["accumulation"].mean()) < 0.999, "vacuous"


@pytest.mark.parametrize("precision", ["fp16x2", "fp32"])
@pytest.mark.parametrize("proposals", [False, True])
def test_uniform_initial_sampler(gpu, precision, proposals):
    """far_plane 6 so that samples uniform in t (spacing 6 / S) resolve the scene the cameras orbit at radius ~1."""
    kw = dict(far_plane=6.0, proposal_initial_sampler="uniform", precision=precision)
    cfg = small_config(num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=24, **kw) if proposals else \
        small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=48, **kw)
    model, out, ref = _pair(cfg, gpu, 45, 52, cam=2, focal=60.0, density_bias=2.0)
    _check(out, ref)
    if proposals:
        for i in (0, 1):
            assert rmse(out[f"prop_depth_{i}"], ref[f"prop_depth_{i}"]) <= TOL
    # ... and it is not the default sampler's image
    cfg_pw = dataclasses.replace(cfg, proposal_initial_sampler="piecewise")
    _, out_pw, _ = _pair(cfg_pw, gpu, 45, 52, cam=2, focal=60.0, density_bias=2.0)
    assert rmse(out["depth"], out_pw["depth"]) > 1e-2


def test_uniform_initial_sampler_with_render_box_and_normals(gpu):
    """Per-ray nears / fars (render_aabb) through the uniform spacing, the small-frame split-depth tail, and the normals kernel."""
    cfg = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=40, proposal_initial_sampler="uniform")
    box = SceneBox(aabb=torch.tensor([[-0.25, -0.2, -0.2], [0.2, 0.25, 0.2]]))
    model, out, ref = _pair(cfg, gpu, 40, 48, cam=1, focal=40.0, render_aabb=box, normals=True)
    hit = ref["depth"] < 1e6
    assert 0.1 < float(hit.float().mean()) < 1.0
    hg = hit.to(gpu)
    for k, c in (("rgb", 3), ("accumulation", 1), ("depth", 1), ("normals", 3), ("pred_normals", 3)):
        e = rmse(out[k][hg.expand(-1, -1, c)], ref[k][hit.expand(-1, -1, c)])
        print(f"{k}: rmse {e:.2e}")
        assert e <= TOL, k


@pytest.mark.parametrize("precision", ["fp16x2", "fp32"])
@pytest.mark.parametrize("proposals", [False, True])
def test_disable_scene_contraction(gpu, precision, proposals):
    """Positions normalised with the model's scene box; samples outside it are dropped by the (0, 1) selector (density 0)."""
    kw = dict(far_plane=4.0, disable_scene_contraction=True, precision=precision)
    cfg = small_config(num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=24, **kw) if proposals else \
        small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=48, **kw)
    model, out, ref = _pair(cfg, gpu, 44, 50, cam=5, focal=55.0, scene_box=SceneBox(aabb=BOX), density_bias=2.0)
    _check(out, ref)
    # ... the box matters: the default unit box gives another image, and so does the contraction
    _, out_unit, ref_unit = _pair(cfg, gpu, 44, 50, cam=5, focal=55.0, density_bias=2.0)
    assert rmse(out_unit["rgb"], ref_unit["rgb"]) <= TOL and rmse(out_unit["rgb"], out["rgb"]) > 1e-3   # (parity errors are ~1e-7)
    _, out_c, _ = _pair(dataclasses.replace(cfg, disable_scene_contraction=False), gpu, 44, 50, cam=5, focal=55.0, density_bias=2.0)
    assert rmse(out_c["rgb"], out["rgb"]) > 1e-3


@pytest.mark.parametrize("precision", ["fp16x2", "fp32"])
def test_disable_scene_contraction_normals(gpu, precision):
    """Field.get_normals differentiates w.r.t. the NORMALISED positions, whichever map produced them; the normals kernel takes the strict
    (IEEE-division) form of the box map."""
    cfg = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=32, far_plane=4.0, disable_scene_contraction=True, precision=precision)
    model, out, ref = _pair(cfg, gpu, 40, 44, cam=3, focal=44.0, scene_box=SceneBox(aabb=BOX), normals=True, density_bias=2.0)
    for k in ("normals", "pred_normals", "rgb"):
        e = rmse(out[k], ref[k])
        print(f"{k}: rmse {e:.2e}")
        assert e <= TOL, k
    assert float(ref["normals"].std()) > 0.05


def test_both_options_with_tcnn_grid(gpu):
    """The tiny-cuda-nn grid semantics go through the same generic instantiations (oracle/tcnn_layout.py: unpinned, as in test_gpu_tcnn).
    The synthetic tiny-cuda-nn checkpoint is an ill-conditioned scene: the ORACLE's own image moves by rmse 6.4e-4 (max 2.5e-2) when
    every normalised position is shifted by one ulp (1.9e-4 with the contraction; 3e-6 for the torch-grid scene) -- measured r03 -- so the
    4.8e-4 seen here is what the fused kernels' reciprocal-form position arithmetic may cost, and the gate is the suite's 1e-3."""
    from helpers import oracle_params_from_tcnn, synthetic_tcnn_checkpoint

    cfg = small_config(num_proposal_samples_per_ray=(48, 24), num_nerf_samples_per_ray=16, far_plane=4.0, implementation="tcnn",
                       proposal_initial_sampler="uniform", disable_scene_contraction=True, average_init_density=3.0)
    sd = synthetic_tcnn_checkpoint(cfg, seed=0)
    model = cfg.setup(scene_box=SceneBox(aabb=BOX))
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    model = model.to(gpu).eval()
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], 44.0, 44.0, 20.0, 18.0, 40, 36).to(gpu)
    bundle = cams[6].generate_rays(camera_indices=0)
    out = model.get_outputs_for_camera_ray_bundle(bundle)
    ref = onf.get_outputs_for_camera_ray_bundle(oracle_params_from_tcnn(sd, cfg), oracle_config(cfg, scene_aabb=BOX.tolist()),
                                                bundle.origins.cpu(), bundle.directions.cpu())
    for k in ("rgb", "accumulation", "depth"):
        e = rmse(out[k], ref[k])
        print(f"{k}: rmse {e:.2e}")
        assert e <= TOL, k
    assert float(ref["rgb"].std()) > 0.03


def test_field_stage_entry_points_follow_the_position_map(gpu):
    """sn_field_forward (the stage-level density / colour entry point the field tests use) with the box map."""
    from signerf_amd import ops

    cfg = small_config(disable_scene_contraction=True)
    model, sd = _model(cfg, gpu, SceneBox(aabb=BOX))
    g = torch.Generator().manual_seed(4)
    pos = (torch.rand(4096, 3, generator=g) * 3.2 - 1.6)           # inside and outside the box
    ocfg = oracle_config(cfg, scene_aabb=BOX.tolist())
    aabb = onf.scene_aabb(ocfg)
    rd, _, q, sel = onf.density_field(sd, "field.mlp_base", ocfg.main, pos[:, None, :], ocfg.average_init_density, aabb)
    inside = sel.view(-1)
    assert 0.2 < float(inside.float().mean()) < 0.8
    for which in (-1, 0, 1):
        if which >= 0:
            rd, _, _, _ = onf.density_field(sd, f"proposal_networks.{which}.mlp_base", ocfg.proposals[which], pos[:, None, :], ocfg.average_init_density, aabb)
        d = ops.field_forward(model, pos.to(gpu), None, which)[0].view(-1).cpu()
        assert float(d[~inside].abs().max()) == 0.0                # outside the box: selector 0
        rel = ((d - rd.view(-1)).abs() / rd.view(-1).clamp_min(1e-6))[inside]
        assert float(rel.max()) < 1e-4, (which, float(rel.max()))


def test_rejected_values(gpu):
    """A scene box without volume, and option values outside {0, 1}, are refused with a message."""
    cfg = small_config(disable_scene_contraction=True)
    model, _ = _model(cfg, gpu, SceneBox(aabb=torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 1.0]])))
    b = Cameras(scene.benchmark_cameras(8)[:, :3], 16.0, 16.0, 8.0, 8.0, 16, 16).to(gpu)[0].generate_rays(0)
    with pytest.raises(_lib.SignerfHipError, match="scene box"):
        model.get_outputs_for_camera_ray_bundle(b)


@pytest.mark.parametrize("proposals", [False, True])
def test_far_plane_beyond_1e7(gpu, proposals):
    """Out there (|p| > ~1.7e7) the exact contraction rounds onto the face q = 1, which the selector drops; the production kernels'
    reciprocal-form position arithmetic may land one ulp inside, so such a far plane is rendered by the strict generic instantiations
    (sn_api.hip needs_generic_kernels).  A thin medium, so that the far samples carry weight."""
    kw = dict(far_plane=1.0e9, background_color="white")
    cfg = small_config(num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=24, **kw) if proposals else \
        small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=48, **kw)
    model, out, ref = _pair(cfg, gpu, 40, 44, cam=4, focal=50.0, density_bias=-2.0)
    for k in ("rgb", "accumulation"):
        e = rmse(out[k], ref[k])
        print(f"{k}: rmse {e:.2e}")
        assert e <= TOL, k
    fin = torch.isfinite(ref["depth"])                         # (a medium this thin may never reach half its weight: median depth inf, both sides)
    assert torch.equal(torch.isfinite(out["depth"].cpu()), fin)
    if bool(fin.any()):
        rel = ((out["depth"].cpu()[fin] - ref["depth"][fin]).abs() / ref["depth"][fin].abs().clamp_min(1.0))
        print(f"depth: max relative error {float(rel.max()):.2e}, median depth range {float(ref['depth'][fin].min()):.3g} .. {float(ref['depth'][fin].max()):.3g}")
        assert float((rel > 1e-3).float().mean()) <= 0.002 and float(rel.median()) <= 1e-5
    assert 0.02 < float(ref["accumulation"].mean()) < 0.98
