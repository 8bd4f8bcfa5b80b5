"""nerfstudio's ``Field`` call surface on the model's Field objects (north_star: "keeping the nerfstudio Model/Field plugin surface";
/root/reference/signerf/signerf.py:27 subclasses NerfactoModel and inherits ``model.field`` / ``model.proposal_networks`` with
``get_density`` / ``density_fn`` / ``get_outputs`` / ``forward``): shapes, dict keys and values against the oracle's field functions."""
import pytest
import torch

from helpers import make_model, oracle_config, small_config
from oracle import nerfacto as onf
from signerf_amd import FieldHeadNames, Frustums, RaySamples, _lib, scene
from signerf_amd.nerfacto import HashMLPDensityField, NerfactoField

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_full(gpu):
    cfg = scene.proposal_config()
    model, sd = make_model(cfg, gpu)
    return cfg, model, sd


def _samples(gpu, R=37, N=11, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(R, 1, 3, generator=g) - 0.5).expand(R, N, 3)
    d = torch.nn.functional.normalize(torch.randn(R, 1, 3, generator=g), dim=-1).expand(R, N, 3)
    bins = torch.cumsum(torch.rand(R, N + 1, 1, generator=g) * 0.4, dim=1)
    bins[:3] *= 30.0                                    # a few rays far outside the unit box: contraction branch
    fr = Frustums(o.contiguous(), d.contiguous(), bins[:, :-1], bins[:, 1:], torch.ones(R, N, 1))
    to = lambda t: t.to(gpu)  # noqa: E731
    return fr, RaySamples(Frustums(to(fr.origins), to(fr.directions), to(fr.starts), to(fr.ends), to(fr.pixel_area)))


@pytest.mark.parametrize("precision", ["fp32", "fp16x2"])
def test_main_field_get_density_get_outputs_forward(model_full, gpu, precision):
    cfg, model, sd = model_full
    model.config.precision = precision
    ocfg = oracle_config(cfg)
    fr, rs = _samples(gpu)
    pos = fr.get_positions()
    rd, rh, _, _ = onf.density_field(sd, "field.mlp_base", ocfg.main, pos, ocfg.average_init_density)
    rrgb = onf.field_rgb(sd, ocfg, fr.directions[:, 0], rh)
    density, emb = model.field.get_density(rs)
    assert density.shape == (37, 11, 1) and emb.shape == (37, 11, 15) and density.device.type == "cuda"
    assert float(((density.cpu() - rd).abs() / rd.clamp_min(1e-6)).max()) <= 1e-4
    scale = float(rh[..., 1:].abs().max())
    assert float((emb.cpu() - rh[..., 1:]).abs().max()) <= 2e-5 * max(scale, 1.0)     # base_mlp_out = layer-2 outputs 1..15
    out = model.field.get_outputs(rs, density_embedding=emb)
    assert list(out.keys()) == [FieldHeadNames.RGB] and out[FieldHeadNames.RGB].shape == (37, 11, 3)
    assert float((out[FieldHeadNames.RGB].cpu() - rrgb).abs().max()) <= 2e-5
    fwd = model.field(rs)                                                             # nn.Module.__call__ -> forward
    assert set(fwd.keys()) == {FieldHeadNames.DENSITY, FieldHeadNames.RGB}
    assert torch.equal(fwd[FieldHeadNames.DENSITY], density) and torch.equal(fwd[FieldHeadNames.RGB], out[FieldHeadNames.RGB])
    assert FieldHeadNames.RGB.value == "rgb" and FieldHeadNames.DENSITY.value == "density"
    with pytest.raises(ValueError):
        model.field.get_outputs(rs, density_embedding=emb[:, :3])
    with pytest.raises(NotImplementedError):
        model.field(rs, compute_normals=True)
    model.config.precision = "fp16x2"


def test_density_fn_of_field_and_proposal_networks(model_full, gpu):
    cfg, model, sd = model_full
    ocfg = oracle_config(cfg)
    g = torch.Generator().manual_seed(3)
    pos = (torch.rand(5, 7, 3, generator=g) - 0.5) * 4.0
    d = model.field.density_fn(pos.to(gpu))
    rd, _, _, _ = onf.density_field(sd, "field.mlp_base", ocfg.main, pos, ocfg.average_init_density)
    assert d.shape == (5, 7, 1) and float(((d.cpu() - rd).abs() / rd.clamp_min(1e-6)).max()) <= 1e-4
    assert len(model.density_fns) == 2
    for i, fn in enumerate(model.density_fns):                                        # what nerfstudio's ProposalNetworkSampler is handed
        di = fn(pos.to(gpu))
        ri, _, _, _ = onf.density_field(sd, f"proposal_networks.{i}.mlp_base", ocfg.proposals[i], pos, ocfg.average_init_density)
        assert di.shape == (5, 7, 1) and float(((di.cpu() - ri).abs() / ri.clamp_min(1e-6)).max()) <= 1e-4
        dens, none = model.proposal_networks[i].get_density(_samples(gpu)[1])
        assert none is None and dens.shape == (37, 11, 1)
    # empty input, and positions on the CPU are moved (the samplers build them wherever the bundle lives)
    assert model.field.density_fn(torch.empty(0, 3, device=gpu)).shape == (0, 1)
    assert torch.equal(model.field.density_fn(pos), d)


def test_a_field_without_a_model_raises(gpu):
    cfg = small_config()
    with pytest.raises(_lib.SignerfHipError, match="not attached"):
        NerfactoField(cfg, 10).density_fn(torch.zeros(4, 3))
    with pytest.raises(_lib.SignerfHipError, match="not attached"):
        HashMLPDensityField().density_fn(torch.zeros(4, 3))


def test_fields_stay_attached_through_reference_style_subclass(gpu):
    """signerf.py:32-39: `populate_modules` calls super() and the model is moved / re-loaded afterwards."""
    from signerf_amd import SIGNeRFModel

    class Sub(SIGNeRFModel):
        def populate_modules(self):
            super().populate_modules()
            self.extra = torch.nn.Linear(2, 2)

    cfg = small_config()
    sd = scene.synthetic_state_dict(cfg, seed=0)
    m = Sub(cfg).to(gpu)
    m.load_state_dict(sd, strict=False)
    pos = torch.rand(9, 3) - 0.5
    ocfg = oracle_config(cfg)
    rd, _, _, _ = onf.density_field(sd, "field.mlp_base", ocfg.main, pos[:, None], ocfg.average_init_density)
    assert float(((m.field.density_fn(pos).cpu() - rd[:, 0]).abs() / rd[:, 0].clamp_min(1e-6)).max()) <= 1e-4
    assert "field._owner_ref" not in m.state_dict() and not any("owner" in k for k in m.state_dict())
