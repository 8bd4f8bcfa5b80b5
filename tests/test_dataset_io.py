"""output formats.  CPU: the transforms reader against the golden fixture produced by the reference's own
load_previous_experiment_cameras; GPU: tensor_to_image's truncating conversion against golden vectors, and a full
write -> read round trip of the dataset directory."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from signerf_amd import dataset_io


def test_load_previous_experiment_cameras_matches_reference(tmp_path):
    with open(os.path.join(GOLDEN, "transforms_roundtrip.json")) as f:
        g = json.load(f)
    p = tmp_path / "transforms.json"
    p.write_text(json.dumps(g["transforms"]))
    ref, syn, combined = dataset_io.load_previous_experiment_cameras(p)
    assert np.array_equal(ref.numpy(), np.array(g["reference_c2w"], dtype=np.float32))
    assert np.array_equal(syn.numpy(), np.array(g["synthetic_c2w"], dtype=np.float32))
    assert combined == g["is_combined"] and ref.shape == (2, 3, 4)


@pytest.mark.gpu
def test_tensor_to_image_truncation_golden(gpu):
    t = np.load(os.path.join(GOLDEN, "tensor_to_image.npz"))
    rgb = dataset_io.tensor_to_image(torch.tensor(t["rgb_in"]).to(gpu))
    gray = dataset_io.tensor_to_image(torch.tensor(t["gray_in"]).to(gpu))
    assert rgb.mode == "RGB" and gray.mode == "L"
    assert np.array_equal(np.array(rgb), t["rgb_out"]) and np.array_equal(np.array(gray), t["gray_out"])   # bit-exact bytes
    back = dataset_io.image_to_tensor(rgb)
    assert np.array_equal(back.numpy(), t["back"])
    # large random image against numpy's own astype
    g = torch.Generator().manual_seed(0)
    x = torch.rand(257, 123, 3, generator=g)
    assert np.array_equal(dataset_io.tensor_to_uint8(x.to(gpu)).cpu().numpy(), (x.numpy() * 255).astype(np.uint8))


@pytest.mark.gpu
def test_dataset_directory_round_trip(gpu, tmp_path):
    from PIL import Image
    from signerf_amd import Cameras, scene

    ds = dataset_io.GeneratedDataset(tmp_path, "experiment-test", downscale_factor=2)
    ds.init_directory()
    for d in ("images", "masks", "conditions", "rendered", "originals", "images_2", "masks_2", "conditions_2", "rendered_2",
              "originals_2", "references"):
        assert (tmp_path / "experiment-test" / d).is_dir()
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], 50.0, 52.0, 16.0, 12.0, 32, 24).to(gpu)
    tr = ds.new_transforms(torch.eye(4)[:3], 0.5, is_synthetic=True)
    g = torch.Generator().manual_seed(3)
    for i in range(3):
        imgs = {"edited": torch.rand(24, 32, 3, generator=g).to(gpu), "render": torch.rand(24, 32, 3, generator=g).to(gpu),
                "mask": (torch.rand(24, 32, 1, generator=g) > 0.5).float().to(gpu), "condition": torch.rand(24, 32, 1, generator=g).to(gpu),
                "edited_scaled": torch.rand(12, 16, 3, generator=g).to(gpu)}
        tr = ds.save_generated_images(i, imgs, cams[i], tr)
        got = np.array(Image.open(tmp_path / "experiment-test" / "images" / f"image_{i}.png"))
        assert np.array_equal(got, (imgs["edited"].cpu().numpy() * 255).astype(np.uint8))
        assert np.array(Image.open(tmp_path / "experiment-test" / "masks" / f"mask_{i}.png")).max() == 255
    tr["reference_indices"], tr["generated_indices"] = [0], [1, 2]
    ds.write_transforms(tr)
    ref, syn, combined = dataset_io.load_previous_experiment_cameras(ds.transforms_path)
    assert torch.equal(ref[0], scene.benchmark_cameras(8)[0, :3]) and syn.shape == (2, 3, 4) and combined is False
    frame = json.load(open(ds.transforms_path))["frames"][1]
    assert set(frame) == {"fl_x", "fl_y", "cx", "cy", "w", "h", "file_path", "_mask_path", "transform_matrix", "scene_transform_matrix"}
    assert frame["fl_y"] == 52.0 and frame["w"] == 32 and frame["file_path"] == "./images/image_1.png"


@pytest.mark.gpu
def test_png_compress_level_changes_bytes_not_pixels(gpu, tmp_path):
    """GeneratedDataset(png_compress_level=1): the faster encoding of the bench's third leg holds the pixels the reference's default call
    writes; written from the thread pool or inline."""
    from PIL import Image

    g = torch.Generator().manual_seed(5)
    x = (torch.rand(96, 128, 3, generator=g) * 0.2 + torch.linspace(0, 0.8, 128)[None, :, None]).to(gpu)
    m = (torch.rand(96, 128, 1, generator=g) > 0.5).float().to(gpu)
    out = {}
    for name, kw in (("default", {}), ("fast", {"png_compress_level": 1, "save_workers": 4})):
        ds = dataset_io.GeneratedDataset(tmp_path, name, **kw)
        ds.init_directory()
        ds.save_image(x, ds.dirs["images"] / "a.png")
        ds.save_image(m, ds.dirs["masks"] / "m.png")
        ds.flush()
        out[name] = [(np.array(Image.open(p)), os.path.getsize(p)) for p in (ds.dirs["images"] / "a.png", ds.dirs["masks"] / "m.png")]
    for (a, na), (b, nb) in zip(out["default"], out["fast"]):
        assert np.array_equal(a, b)
    assert out["fast"][0][1] != out["default"][0][1]
    assert np.array_equal(out["default"][0][0], (x.cpu().numpy() * 255).astype(np.uint8))


def test_native_png_encoder_round_trips_through_pillow(tmp_path):
    """dataset_io.encode_png (the GIL-free encoder of the generator's PNG pool): Pillow decodes exactly the pixels that went in -- RGB and
    greyscale, one-row / one-column images, every compression level the generator exposes."""
    import numpy as np
    from PIL import Image

    from signerf_amd.dataset_io import encode_png

    rng = np.random.default_rng(0)
    for shape in ((37, 53, 3), (64, 64, 1), (1, 17, 3), (23, 1, 1), (2, 2, 3)):
        smooth = (np.add.outer(np.arange(shape[0]), np.arange(shape[1]))[..., None] * 3 + rng.integers(0, 9, shape)).astype(np.uint8)
        for img in (smooth, rng.integers(0, 256, shape, dtype=np.uint8), np.zeros(shape, np.uint8), np.full(shape, 255, np.uint8)):
            for level in (1, 6):
                p = tmp_path / "x.png"
                p.write_bytes(encode_png(img, level))
                back = np.array(Image.open(p))
                assert back.dtype == np.uint8 and np.array_equal(back.reshape(shape), img)
                assert Image.open(p).mode == ("L" if shape[2] == 1 else "RGB")


@pytest.mark.gpu
def test_generated_dataset_writes_the_same_pixels_with_both_encoders(tmp_path, gpu):
    import numpy as np
    from PIL import Image

    t = torch.rand(20, 24, 3, device=gpu)
    m = (torch.rand(20, 24, 1, device=gpu) > 0.5).float()
    out = {}
    for enc in ("native", "pil"):
        ds = dataset_io.GeneratedDataset(tmp_path, enc, save_workers=2, png_encoder=enc)
        ds.init_directory()
        ds.save_image(t, ds.dirs["images"] / "a.png")
        ds.save_image(m, ds.dirs["masks"] / "m.png")
        ds.close()
        out[enc] = (np.array(Image.open(ds.dirs["images"] / "a.png")), np.array(Image.open(ds.dirs["masks"] / "m.png")))
    assert np.array_equal(out["native"][0], out["pil"][0]) and np.array_equal(out["native"][1], out["pil"][1])
    with pytest.raises(ValueError):
        dataset_io.GeneratedDataset(tmp_path, "x", png_encoder="fast")
