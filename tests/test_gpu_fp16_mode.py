"""The opt-in single-fp16 mode of the main field for tiny-cuda-nn checkpoints (``precision="fp16"``, SnRenderOpts.precision = 2; VERDICT r03
item 8): what a real SIGNeRF GPU run computes -- `ns-train nerfacto` checkpoints are trained through tiny-cuda-nn's fp16 FullyFusedMLP
(/root/reference/README.md:146,170) -- against (i) the oracle's emulation of exactly the kernel's roundings (fp16 weights and layer inputs, fp32
accumulation inside a layer, fp16 layer outputs; oracle/nerfacto.py mlp_forward(half=True)) and (ii) the fp32-grade render of the same checkpoint.
Never the default and never the headline: the parity target of the render path is nerfstudio's torch fallback in fp32."""
import pytest
import torch

from helpers import oracle_config, oracle_params_from_tcnn, rmse, small_config, synthetic_tcnn_checkpoint
from oracle import nerfacto as onf
from signerf_amd import Cameras, _lib, ops, scene

pytestmark = pytest.mark.gpu


def _tcnn_model(gpu, seed=0, **kw):
    kw.setdefault("average_init_density", 3.0)
    cfg = small_config(implementation="tcnn", **kw)
    sd = synthetic_tcnn_checkpoint(cfg, seed=seed)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    return cfg, sd, model.to(gpu).eval()


def test_fp16_field_matches_the_emulated_roundings(gpu):
    cfg, sd, model = _tcnn_model(gpu, precision="fp16")
    assert model.effective_precision in ("fp16", "fp16x2")   # (before the first render: the request)
    params = oracle_params_from_tcnn(sd, cfg)
    ocfg = oracle_config(cfg)
    assert ocfg.mlp_precision == "fp16"
    g = torch.Generator().manual_seed(3)
    pos = torch.rand(4096, 3, generator=g) * 2 - 1
    pos[:256] *= 4.0
    d = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1)
    dens, rgb, geo = ops.field_forward(model, pos.to(gpu), d.to(gpu), return_geo=True)
    assert model.effective_precision == "fp16"
    o_d, h, _, _ = onf.density_field(params, "field.mlp_base", ocfg.main, pos[:, None, :], ocfg.average_init_density, half=True)
    o_rgb = onf.field_rgb(params, ocfg, d, h)[:, 0]
    # every layer output is an fp16 value; the kernel's and torch's fp32 summation orders differ in the last bits, so a pre-rounding value
    # that sits within ~1e-7 of a rounding boundary lands on the neighbouring fp16 (2^-11 relative) in one of the two: rare, counted
    rel = (dens.cpu() - o_d[:, 0, 0]).abs() / o_d[:, 0, 0].clamp_min(1e-6)
    print(f"fp16 field vs emulation: density rel p50 {float(rel.median()):.1e} p99 {float(rel.quantile(0.99)):.1e} max {float(rel.max()):.1e}; "
          f"rgb max {float((rgb.cpu() - o_rgb).abs().max()):.1e} rmse {rmse(rgb, o_rgb):.1e}")
    assert float(rel.quantile(0.99)) <= 2e-3 and float(rel.median()) <= 1e-5
    assert rmse(rgb, o_rgb) <= 2e-4 and float((rgb.cpu() - o_rgb).abs().max()) <= 5e-3
    # the geometry features leave the density MLP as fp16 values
    # (the kernel rounds the layer's SCALED output -- an exact power of two, undone afterwards -- so a value that lands in fp16's subnormal range
    #  after un-scaling carries more bits than an unscaled fp16 would: exempt)
    not_h = geo != geo.to(torch.float16).to(torch.float32)
    assert not bool((not_h & (geo.abs() >= 2.0 ** -14)).any()), f"{int((not_h & (geo.abs() >= 2.0 ** -14)).sum())} normal-range geo values are not fp16 values"
    assert float(((geo.cpu() - h[:, 0, 1:]).abs() > 1e-6 * h[:, 0, 1:].abs().clamp_min(1.0)).float().mean()) <= 2e-2
    # and the mode is NOT fp32-grade: it differs from the fp32 evaluation of the same checkpoint at the 1e-3 level
    f_d, fh, _, _ = onf.density_field(params, "field.mlp_base", ocfg.main, pos[:, None, :], ocfg.average_init_density)
    f_rgb = onf.field_rgb(params, oracle_config(small_config(implementation="tcnn", average_init_density=3.0)), d, fh)[:, 0]
    print(f"fp16 field vs the fp32 evaluation: rgb rmse {rmse(rgb, f_rgb):.1e}, density rel p50 "
          f"{float(((dens.cpu() - f_d[:, 0, 0]).abs() / f_d[:, 0, 0].clamp_min(1e-6)).median()):.1e}")
    assert 1e-5 <= rmse(rgb, f_rgb) <= 2e-2


@pytest.mark.parametrize("props", [0, 2])
def test_fp16_render_vs_emulation_and_vs_fp32_grade(gpu, props):
    kw = dict(num_proposal_iterations=props, num_proposal_samples_per_ray=(48, 24) if props else (), num_nerf_samples_per_ray=16)
    cfg, sd, model = _tcnn_model(gpu, precision="fp16", **kw)
    params = oracle_params_from_tcnn(sd, cfg)
    H, W = 40, 56
    b = Cameras(scene.benchmark_cameras(8)[:, :3], 60.0, 60.0, W / 2, H / 2, W, H).to(gpu)[2].generate_rays(0)
    out = {k: v.clone() for k, v in model.get_outputs_for_camera_ray_bundle(b).items() if k in ("rgb", "depth", "accumulation")}
    ref = onf.get_outputs_for_camera_ray_bundle(params, oracle_config(cfg), b.origins.cpu(), b.directions.cpu())
    e = {k: rmse(out[k], ref[k]) for k in out}
    model.config.precision = "fp16x2"
    grade = model.get_outputs_for_camera_ray_bundle(b)
    moved = {k: rmse(out[k], grade[k]) for k in out}
    print(f"fp16 render props={props}: vs emulation {{{', '.join(f'{k} {v:.1e}' for k, v in e.items())}}}; "
          f"moved from the fp32-grade render by {{{', '.join(f'{k} {v:.1e}' for k, v in moved.items())}}}")
    assert e["rgb"] <= 3e-4 and e["accumulation"] <= 3e-4 and e["depth"] <= 1e-2     # (a median-depth flip is a whole bin)
    assert 1e-6 <= moved["rgb"] <= 1e-2                                               # a different arithmetic, at the 1e-3 level
    assert float(ref["rgb"].std()) > 0.05


def test_fp16_storage_of_the_grid_equals_rounding_on_the_fly(gpu, monkeypatch):
    """The mode's grid values are the table rounded through fp16 once (what tiny-cuda-nn's `params.to(half)` holds).  Two ways to read them: the
    fp16 STORAGE (SnFieldDesc.half_grid: 16-byte quads of the de-hashed levels, 4-byte rows of the hashed ones -- 62 gathers per sample instead
    of 84) and the uploaded fp32 table with every row rounded on the fly (a handle without the storage).  Same values, same blend: bit-identical."""
    kw = dict(num_proposal_iterations=0, num_nerf_samples_per_ray=24, precision="fp16")
    full = dict(log2_hashmap_size=19)       # nerfacto's table size: 5 densely indexed levels + 6 more de-hashed ones, 5 hashed
    H, W = 48, 40
    outs, layouts = [], []
    for env in (None, "0"):
        if env is None:
            monkeypatch.delenv("SN_HALF_GRID", raising=False)
        else:
            monkeypatch.setenv("SN_HALF_GRID", env)
        cfg, sd, model = _tcnn_model(gpu, seed=5, **kw, **full)
        b = Cameras(scene.benchmark_cameras(8)[:, :3], 55.0, 55.0, W / 2, H / 2, W, H).to(gpu)[1].generate_rays(0)
        outs.append({k: v.clone() for k, v in model.get_outputs_for_camera_ray_bundle(b).items() if k in ("rgb", "depth", "accumulation", "expected_depth")})
        layouts.append(ops.debug_layout(model))
        assert model.effective_precision == "fp16"
    assert layouts[0]["half_grid_bytes"] > 0 and layouts[1]["half_grid_bytes"] == 0
    assert layouts[0]["n_dense"] == 11
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    # and the storage is what the layout says: 16 bytes per grid point of the copied levels + 4 bytes per row of the others
    # (+ the five hashed levels as fp16 x-pairs: 8 bytes per entry, one table per count of trailing one bits of the x coordinate -- 10 to 14 per level)
    quads = sum((r ** 3 * 16 + 255) // 256 * 256 for r in layouts[0]["dense_res"][:11])
    assert quads + 5 * 10 * (1 << 19) * 8 <= layouts[0]["half_grid_bytes"] <= quads + 5 * 14 * (1 << 19) * 8


def test_switching_the_precision_on_later_rebuilds_the_handle_with_the_storage(gpu):
    """`model.config.precision = "fp16"` on a model that has already rendered at fp32 grade: the handle is re-created once with the grid's
    fp16 storage (the fast path), and the render equals that of a model configured so from the start."""
    kw = dict(num_proposal_iterations=0, num_nerf_samples_per_ray=16)
    cfg, sd, model = _tcnn_model(gpu, seed=1, precision="fp16x2", **kw)
    b = Cameras(scene.benchmark_cameras(8)[:, :3], 50.0, 50.0, 16.0, 12.0, 32, 24).to(gpu)[3].generate_rays(0)
    model.get_outputs_for_camera_ray_bundle(b)
    assert ops.debug_layout(model)["half_grid_bytes"] == 0
    model.config.precision = "fp16"
    late = model.get_outputs_for_camera_ray_bundle(b)["rgb"].clone()
    assert ops.debug_layout(model)["half_grid_bytes"] > 0 and model.effective_precision == "fp16"
    _, _, ref = _tcnn_model(gpu, seed=1, precision="fp16", **kw)
    assert torch.equal(late, ref.get_outputs_for_camera_ray_bundle(b)["rgb"])
    model.config.precision = "fp16x2"                      # ... and back: the same handle, the storage simply stays
    model.get_outputs_for_camera_ray_bundle(b)
    assert ops.debug_layout(model)["half_grid_bytes"] > 0


def test_fp16_mode_is_for_tcnn_checkpoints_only(gpu):
    with pytest.raises(NotImplementedError, match="tcnn"):
        small_config(precision="fp16").setup()                                        # implementation="torch": the parity target stays fp32-grade
    # ... and the C ABI says the same when asked directly
    cfg = small_config()
    model = cfg.setup()
    model.load_state_dict(scene.synthetic_state_dict(cfg, seed=0), strict=False)
    model = model.to(gpu).eval()
    lib = model._ensure_engine()
    pos = torch.zeros(8, 3, device=gpu)
    dens = torch.empty(8, device=gpu)
    st = lib.sn_field_forward(model._handle, -1, pos.data_ptr(), None, 8, 2, dens.data_ptr(), None, _lib.current_stream())
    assert st == 1 and b"grid_mode 1" in lib.sn_last_error(model._handle)
    assert lib.sn_effective_precision(model._handle, 2, 0) == -1
