"""SURVEY §8(f) row 1, second half on the GPU: the bilinear down/up-scale and the reference-sheet composition
(datasetgenerator.py:497-593, 597-674) against torch's own F.interpolate on the CPU -- the call the reference makes."""
import pytest
import torch

from helpers import make_model, small_config
from oracle import signerf_utils as su
from signerf_amd import Cameras, scene
from signerf_amd.datasetgenerator import (DatasetGeneratorConfig, cell_window, compose_reference_sheet, generate_reference_sheet,
                                          generate_with_reference_sheet, sheet_geometry)
from signerf_amd.ops import resize_bilinear

pytestmark = pytest.mark.gpu

# float work: PyTorch's CPU kernel may fuse / reorder the four products; 2 ulp of a value in [0, 1]
TOL = 3e-7


def _blob_mask(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cy, cx = (0.3 + 0.4 * torch.rand(2, generator=g)) * torch.tensor([H, W])
    r = 0.15 * min(H, W) * (1 + torch.rand(1, generator=g))
    return (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).unsqueeze(-1)


@pytest.mark.parametrize("H,W,h,w,C", [
    (800, 800, 400, 400, 3),     # the reference's default 1/2 downscale (exact 2x2 averages)
    (1080, 1920, 540, 960, 1),
    (101, 67, 37, 29, 3),        # ragged, non-integer ratio
    (400, 400, 800, 800, 3),     # the up-scale of an edited cell (:586)
    (50, 30, 50, 30, 2),         # identity size
    (1, 1, 4, 4, 1),             # degenerate source
])
def test_resize_matches_torch_interpolate(gpu, H, W, h, w, C):
    g = torch.Generator().manual_seed(H * 31 + w)
    src = torch.rand(H, W, C, generator=g)
    out = resize_bilinear(src.to(gpu), h, w).cpu()
    ref = su.interpolate_hwc(src, h, w)
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) <= TOL


def test_resize_between_windows(gpu):
    """Source and destination given as windows of larger images (row strides): the down-scale IS the paste."""
    g = torch.Generator().manual_seed(5)
    big = torch.rand(120, 90, 3, generator=g)
    sheet = torch.full((64, 80, 3), 7.0)
    d_sheet = sheet.to(gpu)
    resize_bilinear(big.to(gpu)[10:110, 5:85, :], 25, 20, out=d_sheet[8:33, 40:60, :])
    ref = sheet.clone()
    ref[8:33, 40:60, :] = su.interpolate_hwc(big[10:110, 5:85, :].contiguous(), 25, 20)
    assert float((d_sheet.cpu() - ref).abs().max()) <= TOL
    assert torch.equal(d_sheet.cpu()[:8], ref[:8])  # nothing outside the window is touched


@pytest.mark.parametrize("H,W,h,w", [(800, 800, 400, 400), (97, 131, 48, 65), (60, 60, 25, 31)])
def test_mask_downscale_threshold(gpu, H, W, h, w):
    """mask_scaled = interpolate(mask.float()) > 0.5 (:527).  Boolean work: exact, except where the interpolated value is
    within rounding of the threshold (cannot happen for the exact 2x2 averages of the default factor)."""
    mask = _blob_mask(H, W, seed=H + w)
    out = resize_bilinear(mask.to(gpu), h, w, threshold=True).cpu()
    val = su.interpolate_hwc(mask.float(), h, w)
    ref = (val > 0.5).float()
    differ = out != ref
    if (H, W, h, w) == (800, 800, 400, 400):
        assert not bool(differ.any())
    assert bool(((val - 0.5).abs()[differ] <= TOL).all())
    assert set(out.unique().tolist()) <= {0.0, 1.0}
    # uint8 and bool sources are the same thing
    out_u8 = resize_bilinear(mask.to(torch.uint8).to(gpu), h, w, threshold=True).cpu()
    assert torch.equal(out, out_u8)


@pytest.mark.parametrize("rows,cols,H,W,sw,sh,border", [
    (2, 3, 96, 128, 64, 48, 0),   # the reference's 2 x 3 sheet, factor 2
    (2, 2, 90, 70, 33, 41, 3),    # borders, sizes that need the pad to a multiple of 8
    (1, 2, 40, 40, 20, 20, 5),
])
def test_compose_reference_sheet_matches_oracle(gpu, rows, cols, H, W, sw, sh, border):
    cfg = DatasetGeneratorConfig(rows=rows, cols=cols, border_width_between_images=border)
    g = torch.Generator().manual_seed(rows * 100 + cols)
    views = [(torch.rand(H, W, 3, generator=g), _blob_mask(H, W, seed=i), torch.rand(H, W, 1, generator=g)) for i in range(rows * cols - 1)]
    img, msk, cnd, refs = compose_reference_sheet(cfg, [(r.to(gpu), m.to(gpu), c.to(gpu)) for r, m, c in views], sw, sh)
    o_img, o_msk, o_cnd = su.compose_reference_sheet(views, rows, cols, sw, sh, border)
    assert tuple(img.shape[:2]) == sheet_geometry(cfg, sw, sh) == tuple(o_img.shape[:2])
    assert img.shape[0] % 8 == 0 and img.shape[1] % 8 == 0
    assert float((img.cpu() - o_img).abs().max()) <= TOL
    assert float((cnd.cpu() - o_cnd).abs().max()) <= TOL
    differ = msk.cpu() != o_msk
    assert int(differ.sum()) <= 2  # threshold near-ties only (non-integer ratios)
    # the last cell and the borders keep the fill values (ones / zeros)
    r0, r1, c0, c1 = cell_window(cfg, rows * cols - 1, sw, sh)
    assert bool((img[r0:r1, c0:c1] == 1).all()) and bool((msk[r0:r1, c0:c1] == 0).all())
    assert len(refs) == rows * cols - 1 and refs[0]["mask_scaled"].dtype == torch.bool
    with pytest.raises(ValueError):
        compose_reference_sheet(cfg, [(r.to(gpu), m.to(gpu), c.to(gpu)) for r, m, c in views[:-1]] if len(views) > 1 else [], sw, sh)


def test_generate_reference_sheet_end_to_end(gpu):
    """render -> mask -> condition -> sheet -> (stand-in) diffuser -> blend -> cut -> up-scale, all on the GPU, checked against
    the same steps done with torch on the CPU from the GPU renders."""
    cfg_m = small_config(num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=12)
    model, _ = make_model(cfg_m, gpu, density_bias=5.0)
    H = W = 64
    c2w = scene.benchmark_cameras(8)[:, :3]
    cams = Cameras(c2w, 1.2 * W, 1.2 * W, W / 2, H / 2, W, H).to(gpu)
    cfg = DatasetGeneratorConfig(rows=2, cols=2, mask_dialation=(9, 9), aabb_min=[-0.3, -0.3, -0.3], aabb_max=[0.3, 0.3, 0.3])
    seen = {}

    def diffuse(image, image2, mask, cond):
        seen["shapes"] = (tuple(image.shape), tuple(mask.shape), tuple(cond.shape))
        return 1.0 - image  # a deterministic "edit"

    sw = sh = 32
    img, msk, cnd, edited, refs = generate_reference_sheet(cfg, model, [cams[i] for i in range(3)], sw, sh, diffuse)
    assert seen["shapes"] == ((64, 64, 3), (64, 64, 1), (64, 64, 1))
    views = [(r["render"].cpu(), r["mask"].cpu(), r["condition"].cpu()) for r in refs]
    o_img, o_msk, o_cnd = su.compose_reference_sheet(views, 2, 2, sw, sh, 0)
    assert float((img.cpu() - o_img).abs().max()) <= TOL and torch.equal(msk.cpu(), o_msk)
    o_edit = (1.0 - o_img) * o_msk.repeat(1, 1, 3) + o_img * (1 - o_msk.repeat(1, 1, 3))
    assert float((edited.cpu() - o_edit).abs().max()) <= 2 * TOL
    for i, r in enumerate(refs):
        r0, r1, c0, c1 = cell_window(cfg, i, sw, sh)
        up = su.interpolate_hwc(o_edit[r0:r1, c0:c1, :].contiguous(), H, W)
        assert float((r["edited"].cpu() - up).abs().max()) <= 4 * TOL
    # the per-view step that follows (:597-674): the new view goes into the LAST cell
    out = generate_with_reference_sheet(cfg, model, cams[3], None, sw, sh, edited.clone(), cnd.clone(), diffuse)
    assert out["edited"].shape == (H, W, 3) and out["mask_scaled"].dtype == torch.bool
    rs = su.interpolate_hwc(out["render"].cpu(), sh, sw)
    assert float((out["render_scaled"].cpu() - rs).abs().max()) <= TOL
    ms = out["mask_scaled"].cpu()
    e_s = (1.0 - rs) * ms + rs * (~ms)
    assert float((out["edited_scaled"].cpu() - e_s).abs().max()) <= 2 * TOL
