"""``DatasetGenerator.generate_dataset`` (BASELINE.json configs[4]; /root/reference/signerf/datasetgenerator/datasetgenerator.py:185-393)
on the CPU: the ORCHESTRATION -- pre-computed sharded renders, the serial diffusion sequence with the in-place last cell (:643-646),
the directory and transforms.json -- with test-only stand-ins for the three HIP-only pieces (render_camera, the bilinear resize, the
uint8 cast).  The same loop with real renders across two processes is tests/test_gpu_generate_dataset.py."""
import hashlib
import json
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


RENDERED = []   # (per process) the cameras this process rendered


def _fake_render_camera(config, graph, camera, with_mask=True, with_condition=True):
    host = camera._host.reshape(-1)
    seed = int.from_bytes(hashlib.sha1(host.numpy().tobytes()).digest()[:4], "little")
    RENDERED.append(seed)
    g = torch.Generator().manual_seed(seed)
    H, W = int(host[17]), int(host[16])
    rgb = torch.rand(H, W, 3, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    cy, cx = (torch.rand(2, generator=g) * 0.5 + 0.25) * torch.tensor([H, W])
    mask = (((yy - cy) ** 2 + (xx - cx) ** 2) < (0.3 * H) ** 2)[..., None]
    cond = torch.rand(H, W, 1, generator=g) * mask
    return rgb, mask, cond


def _fake_resize(src, out_h, out_w, threshold=False, out=None):
    r = F.interpolate(src.float().permute(2, 0, 1)[None], (out_h, out_w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    if threshold:
        r = (r > 0.5).float()
    if out is not None:
        out.copy_(r)
        return out
    return r.contiguous()


def install_cpu_standins():
    from signerf_amd import dataset_io, datasetgenerator, ops

    datasetgenerator.render_camera = _fake_render_camera
    ops.resize_bilinear = _fake_resize
    dataset_io.tensor_to_uint8 = lambda t: (t.detach().float() * 255).to(torch.uint8)


@pytest.fixture()
def standins(monkeypatch):
    from signerf_amd import dataset_io, datasetgenerator, ops

    monkeypatch.setattr(datasetgenerator, "render_camera", _fake_render_camera)
    monkeypatch.setattr(ops, "resize_bilinear", _fake_resize)
    monkeypatch.setattr(dataset_io, "tensor_to_uint8", lambda t: (t.detach().float() * 255).to(torch.uint8))
    RENDERED.clear()


class _Graph:
    device = torch.device("cpu")
    render_aabb = None


def _poses(n_ref=5, n_views=7):
    from signerf_amd import circle_poses, random_sphere_poses

    ref = circle_poses(n_ref, torch.device("cpu"), 0.5, 90.0, (0.0, 300.0), [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])[:, :3]
    torch.manual_seed(1)
    syn = random_sphere_poses(n_views, torch.device("cpu"), 0.5, (30.0, 120.0), (0.0, 360.0), [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])[:, :3]
    return ref, syn


def _generator(path, name, diffuse=None, **kw):
    from signerf_amd.datasetgenerator import DatasetGenerator, DatasetGeneratorConfig

    cfg = DatasetGeneratorConfig(path=path, dataset_name=name, fx=40.0, fy=40.0, cx=SIZE / 2, cy=SIZE / 2, width=SIZE, height=SIZE,
                                 mask_dialation=(5, 5))
    return DatasetGenerator(cfg, torch.eye(4)[:3], 1.0, None, device="cpu", diffuse=diffuse, **kw)


def _tree(root):
    out = {}
    for d, _, files in os.walk(root):
        for f in files:
            p = os.path.join(d, f)
            out[os.path.relpath(p, root)] = open(p, "rb").read()
    return out


def test_generate_dataset_layout_and_transforms(tmp_path, standins):
    from signerf_amd.dataset_io import load_previous_experiment_cameras

    ref, syn = _poses()
    gen = _generator(tmp_path, "exp")
    gen.generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
    root = tmp_path / "exp"
    t = json.load(open(root / "transforms.json"))
    assert t["method"] == "SIGNeRF" and t["is_synthetic"] is True and t["is_combined"] is False and t["camera_model"] == "OPENCV"
    assert t["reference_indices"] == list(range(5)) and t["generated_indices"] == list(range(5, 12)) and "original_indices" not in t
    assert len(t["frames"]) == 12 and t["frames"][7]["file_path"] == "./images/image_7.png" and t["frames"][7]["_mask_path"] == "./masks/mask_7.png"
    assert t["frames"][0]["w"] == SIZE and t["frames"][0]["fl_x"] == 40.0
    for sub, stem in (("images", "image"), ("rendered", "image"), ("masks", "mask"), ("conditions", "condition"),
                      ("images_2", "image"), ("rendered_2", "image"), ("masks_2", "mask"), ("conditions_2", "condition")):
        assert sorted(os.listdir(root / sub)) == sorted(f"{stem}_{i}.png" for i in range(12)), sub
    assert os.listdir(root / "originals") == [] and os.listdir(root / "originals_2") == []
    assert sorted(os.listdir(root / "references")) == ["condition_reference_sheet.png", "edited_reference_sheet.png",
                                                       "image_reference_sheet.png", "mask_reference_sheet.png"]
    assert (root / "config.yml").exists()
    # the reader of the same file (load_previous_experiment_cameras.py) gets the cameras back
    r, s, combined = load_previous_experiment_cameras(root / "transforms.json")
    assert torch.equal(r, ref) and torch.equal(s, syn) and combined is False
    assert len(RENDERED) == 12 and len(set(RENDERED)) == 12          # every camera rendered exactly once (pre-computed, then looked up)
    from PIL import Image

    assert Image.open(root / "images" / "image_3.png").size == (SIZE, SIZE) and Image.open(root / "images_2" / "image_3.png").size == (SIZE // 2,) * 2
    sheet = Image.open(root / "references" / "image_reference_sheet.png")
    assert sheet.size == (3 * SIZE // 2, 2 * SIZE // 2) and sheet.mode == "RGB"


def test_diffuser_sequence_and_in_place_last_cell(tmp_path, standins):
    """datasetgenerator.py:558 and :643-650: one call for the reference sheet, then one per generated view on the EDITED sheet whose
    last cell holds that view (mutated in place: the same tensor objects every call), mask sheet zero outside the last cell."""
    from signerf_amd.datasetgenerator import cell_window

    calls = []

    def diffuse(original, rendered, mask, condition):
        assert original is rendered
        calls.append((original, original.clone(), mask.clone(), condition.clone(), condition))
        return 1.0 - original   # a visible "edit"

    ref, syn = _poses()
    gen = _generator(tmp_path, "exp", diffuse=diffuse, write_images=False)
    gen.generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
    assert len(calls) == 1 + 7
    cfg, half = gen.config, SIZE // 2
    r0, r1, c0, c1 = cell_window(cfg, 5, half, half)
    first_img, first_mask = calls[0][1], calls[0][2]
    assert torch.equal(first_img[r0:r1, c0:c1], torch.ones(half, half, 3)) and float(first_mask[r0:r1, c0:c1].abs().max()) == 0   # empty last cell
    edited_ref = first_img * (1 - first_mask) + (1 - first_img) * first_mask      # what the sheet stage keeps: the edit inside the mask
    sheets = {id(c[0]) for c in calls[1:]}
    conds = {id(c[4]) for c in calls[1:]}
    assert len(sheets) == 1 and len(conds) == 1                                    # the per-view calls see ONE image sheet / condition sheet object
    from signerf_amd import Cameras

    cams = Cameras(syn, 40.0, 40.0, SIZE / 2, SIZE / 2, SIZE, SIZE)
    for k in range(7):
        _, img, mask, cond, _ = calls[1 + k]
        rgb, m, cnd = _fake_render_camera(cfg, None, cams[k])
        assert torch.equal(img[r0:r1, c0:c1], _fake_resize(rgb, half, half))       # this view in the last cell
        assert torch.equal(mask[r0:r1, c0:c1], _fake_resize(m, half, half, threshold=True))
        assert torch.equal(cond[r0:r1, c0:c1], _fake_resize(cnd, half, half))
        outside = torch.ones_like(mask, dtype=torch.bool)
        outside[r0:r1, c0:c1] = False
        assert float(mask[outside].abs().max()) == 0                               # fresh mask sheet per view (:644)
        for cell in range(5):                                                      # the edited reference cells are what every view sees
            a0, a1, b0, b1 = cell_window(cfg, cell, half, half)
            assert torch.allclose(img[a0:a1, b0:b1], edited_ref[a0:a1, b0:b1], atol=1e-6)
            assert torch.equal(cond[a0:a1, b0:b1], calls[0][3][a0:a1, b0:b1])
    # as the reference leaves them: the sheets hold the LAST view
    assert torch.equal(gen.edited_reference_sheet[r0:r1, c0:c1], calls[-1][1][r0:r1, c0:c1])


def test_precompute_off_writes_the_same_dataset(tmp_path, standins):
    ref, syn = _poses()
    _generator(tmp_path, "a", save_workers=0).generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
    _generator(tmp_path, "b", precompute=False, save_workers=4).generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
    a, b = _tree(tmp_path / "a"), _tree(tmp_path / "b")
    a.pop("config.yml"), b.pop("config.yml")   # (holds the dataset name)
    assert a.keys() == b.keys() and all(a[k] == b[k] for k in a)


def test_argument_errors(tmp_path, standins):
    ref, syn = _poses()
    gen = _generator(tmp_path, "exp")
    with pytest.raises(ValueError, match="Either original dataset or camera_to_worlds"):
        gen.generate_dataset(_Graph(), ref)
    with pytest.raises(ValueError, match="to merge with original dataset"):
        gen.generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn, merge_with_original_dataset=True)
    with pytest.raises(ValueError, match="is not equal to"):
        gen.generate_dataset(_Graph(), ref[:4], synthetic_camera_to_worlds=syn)


class _Original:
    """Duck-typed nerfstudio InputDataset: .cameras, ._dataparser_outputs.image_filenames, .get_image_float32."""

    def __init__(self, cameras, filenames, images):
        self.cameras, self._images = cameras, images
        self._dataparser_outputs = type("O", (), {"image_filenames": filenames})()

    def get_image_float32(self, idx):
        return self._images[idx]


def test_merge_with_original_dataset(tmp_path, standins):
    """:344-388: the original views are appended with INVERTED masks, their renders under originals/, the photo as the image."""
    from PIL import Image

    from signerf_amd import Cameras

    ref, syn = _poses(n_views=3)
    torch.manual_seed(3)
    from signerf_amd import random_sphere_poses

    orig_c2w = random_sphere_poses(2, torch.device("cpu"), 0.6, (40.0, 100.0), (0.0, 360.0), [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])[:, :3]
    photos = [torch.rand(SIZE, SIZE, 3) for _ in range(2)]
    ds = _Original(Cameras(orig_c2w, 40.0, 40.0, SIZE / 2, SIZE / 2, SIZE, SIZE), [None, None], photos)
    gen = _generator(tmp_path, "exp")
    gen.generate_dataset(_Graph(), ref, original_dataset=ds, synthetic_camera_to_worlds=syn, merge_with_original_dataset=True)
    root = tmp_path / "exp"
    t = json.load(open(root / "transforms.json"))
    assert t["is_combined"] is True and t["original_indices"] == [8, 9] and len(t["frames"]) == 10
    assert sorted(os.listdir(root / "originals")) == ["image_8.png", "image_9.png"]
    import numpy as np

    assert len(RENDERED) == 10 and len(set(RENDERED)) == 10      # 5 + 3 + 2 cameras, each rendered once (all pre-computed)
    got = np.array(Image.open(root / "images" / "image_8.png"))
    assert np.array_equal(got, (photos[0] * 255).to(torch.uint8).numpy())
    _, m, _ = _fake_render_camera(None, None, ds.cameras[0])
    assert np.array_equal(np.array(Image.open(root / "masks" / "mask_8.png")) > 0, (~m).squeeze(-1).numpy())


class _ForeignCameras:
    """`original_dataset.cameras` as a real run hands it over (datasetgenerator.py:274-275): a nerfstudio object, NOT this package's class --
    attribute tensors, per-camera intrinsics, OPENCV distortion, integer indexing that yields 0-dim cameras, ``.to`` and ``.size``."""

    def __init__(self, c2w, fx, sizes, dist):
        self.camera_to_worlds, self.fx, self.fy = c2w, fx, fx * 1.02
        self.width, self.height = sizes[:, :1].clone(), sizes[:, 1:].clone()
        self.cx, self.cy = self.width.float() / 2, self.height.float() / 2
        self.distortion_params, self.camera_type = dist, torch.ones(c2w.shape[0], 1, dtype=torch.int64)
        self.times = self.metadata = None

    size = property(lambda self: self.camera_to_worlds.shape[0])

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        one = _ForeignCameras.__new__(_ForeignCameras)
        for k in ("camera_to_worlds", "fx", "fy", "cx", "cy", "width", "height", "distortion_params", "camera_type"):
            setattr(one, k, getattr(self, k)[i])
        one.times = one.metadata = None
        return one

    def to(self, device):
        return self


def _original_dataset(tmp_path, n=6, second_size=None):
    from PIL import Image

    from signerf_amd import random_sphere_poses

    torch.manual_seed(5)
    c2w = random_sphere_poses(n, torch.device("cpu"), 0.6, (40.0, 100.0), (0.0, 360.0), [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])[:, :3]
    sizes = torch.full((n, 2), SIZE, dtype=torch.int64)
    if second_size is not None:
        sizes[n // 2:] = torch.tensor(second_size)
    fx = torch.linspace(38.0, 44.0, n)[:, None]
    dist = torch.tensor([0.05, -0.02, 0.0, 0.0, 0.001, -0.002]).expand(n, 6) * torch.linspace(0.5, 1.5, n)[:, None]
    cams = _ForeignCameras(c2w, fx, sizes, dist)
    files, photos = [], []
    os.makedirs(tmp_path / "photos", exist_ok=True)
    for i in range(n):
        w, h = int(sizes[i, 0]), int(sizes[i, 1])
        photo = (torch.rand(h, w, 3) * 255).to(torch.uint8)
        f = tmp_path / "photos" / f"frame_{i}.png"
        if not f.exists():
            Image.fromarray(photo.numpy()).save(f)
        files.append(f)
        photos.append(photo.float() / 255)
    return _Original(cams, files, photos)


def test_original_cameras_mode_precomputes_foreign_distorted_cameras(tmp_path, standins):
    """The reference's DEFAULT source of the generated views (`cameras = original_dataset.cameras`, :274-275; signerf_trainer.py:222) --
    foreign camera objects with per-camera intrinsics and lens parameters: every view is pre-computed (r03 fell back to the serial
    loop whenever a camera was not this package's class), each camera rendered once, intrinsics and lens carried through."""
    ref, _ = _poses()
    ds = _original_dataset(tmp_path)
    gen = _generator(tmp_path, "exp")
    seen = []
    real = gen.precompute_views
    gen.precompute_views = lambda graph, cams: (seen.append(len(cams)), real(graph, cams))[1]
    gen.generate_dataset(_Graph(), ref, original_dataset=ds)
    assert seen == [5 + 6] and gen.precompute_skipped == 0
    assert len(RENDERED) == 11 and len(set(RENDERED)) == 11
    t = json.load(open(tmp_path / "exp" / "transforms.json"))
    assert t["is_synthetic"] is False and t["generated_indices"] == list(range(5, 11))
    assert abs(t["frames"][5]["fl_x"] - 38.0) < 1e-6 and abs(t["frames"][10]["fl_x"] - 44.0) < 1e-5     # per-camera intrinsics survive the adoption
    assert sorted(os.listdir(tmp_path / "exp" / "originals")) == sorted(f"image_{i}.png" for i in range(5, 11))  # the loaded photo replaces the render (:628-630)


def test_mixed_image_sizes_and_memory_budget(tmp_path, standins):
    """One gather per image size; views beyond the budget are rendered inside the serial loop -- the dataset is the same either way."""
    ref, _ = _poses()
    ds = _original_dataset(tmp_path, second_size=(SIZE + 8, SIZE - 8))
    gen = _generator(tmp_path, "all")
    sizes = []
    real = gen.precompute_views
    gen.precompute_views = lambda graph, cams: (sizes.append((len(cams), int(cams[0].width), int(cams[0].height))), real(graph, cams))[1]
    gen.generate_dataset(_Graph(), ref, original_dataset=ds)
    assert sizes == [(5 + 3, SIZE, SIZE), (3, SIZE + 8, SIZE - 8)]
    RENDERED.clear()
    tight = _generator(tmp_path, "tight", precompute_budget_mb=0)          # nothing fits: everything rendered in the loop
    tight.generate_dataset(_Graph(), ref, original_dataset=ds)
    assert tight.precompute_skipped == 11 and len(RENDERED) == 11
    a, b = _tree(tmp_path / "all"), _tree(tmp_path / "tight")
    a.pop("config.yml"), b.pop("config.yml")
    assert a.keys() == b.keys() and all(a[k] == b[k] for k in a)


def test_writer_threads_are_released(tmp_path, standins):
    import threading

    ref, syn = _poses(n_views=2)
    before = threading.active_count()
    for k in range(3):
        gen = _generator(tmp_path, f"exp{k}", save_workers=4)
        gen.generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
        assert gen.dataset._pool is None and not gen.dataset._pending
    assert threading.active_count() <= before                              # r03 leaked one pool per generate_dataset call (ADVICE)

    def broken(*a):
        raise RuntimeError("diffuser down")

    gen = _generator(tmp_path, "broken", diffuse=broken, save_workers=4)
    with pytest.raises(RuntimeError, match="diffuser down"):
        gen.generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
    assert gen.dataset._pool is None and threading.active_count() <= before


# ---- world 2 over gloo: the written dataset equals the single-process one ----------------------------------------------------------
def _worker(rank, world, port, out_dir, mode="synthetic"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_generate_dataset as me

    me.install_cpu_standins()
    ref, syn = me._poses()
    import pathlib

    if mode == "original":   # the reference's default camera source: a foreign camera batch, distorted, per-camera intrinsics
        gen = me._generator(out_dir, "sharded", serial_stage_timeout_s=120.0)
        gen.generate_dataset(me._Graph(), ref, original_dataset=me._original_dataset(pathlib.Path(out_dir)))
    elif mode == "diffuser_raises":   # rank 0 leaves the serial stage through an exception: the idle ranks must not stay parked
        def broken(*a, **k):
            raise ConnectionError("diffuser down")

        gen = me._generator(out_dir, "sharded", diffuse=broken, serial_stage_timeout_s=600.0)
        t0, raised = time.perf_counter(), None
        try:
            gen.generate_dataset(me._Graph(), ref, synthetic_camera_to_worlds=syn)
        except ConnectionError as e:
            raised = str(e)
        json.dump({"raised": raised, "seconds": time.perf_counter() - t0}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
        dist.destroy_process_group()
        return
    elif mode in ("init_raises", "init_raises_precompute"):
        # rank 0 fails BEFORE the serial stage (an unwritable path).  Nothing pre-computed: the other ranks go straight to the
        # closing barrier and must be met there.  ADVICE r05, pre-compute ON (the default): the other ranks must not be left alone in the
        # stage-1 collectives either -- they learn of the failure through the budget reduce and raise a RuntimeError of their own.
        gen = me._generator(out_dir, "sharded", precompute=(mode == "init_raises_precompute"), serial_stage_timeout_s=600.0)
        if rank == 0:
            def broken_init():
                raise PermissionError("cannot create the dataset directory")

            gen.init_directory = broken_init
        t0, raised = time.perf_counter(), None
        try:
            gen.generate_dataset(me._Graph(), ref, synthetic_camera_to_worlds=syn)
        except PermissionError as e:
            raised = str(e)
        except RuntimeError as e:
            raised = "RuntimeError: " + str(e)
        json.dump({"raised": raised, "seconds": time.perf_counter() - t0, "rendered": me.RENDERED}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
        dist.destroy_process_group()
        return
    else:
        gen = me._generator(out_dir, "sharded", finish_sync=(mode != "nosync"))
        gen.generate_dataset(me._Graph(), ref, synthetic_camera_to_worlds=syn)
    json.dump({"rendered": me.RENDERED, "wrote": os.path.exists(os.path.join(out_dir, "sharded", "transforms.json"))},
              open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_generate_dataset_sharded_over_gloo_equals_single_process(tmp_path, standins, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref, syn = _poses()
    _generator(tmp_path, "single").generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
    a, b = _tree(tmp_path / "sharded"), _tree(tmp_path / "single")
    a.pop("config.yml"), b.pop("config.yml")
    assert a.keys() == b.keys()
    for k in a:
        assert a[k] == b[k], f"{k} differs between the {world}-rank and the single-process dataset"
    per_rank = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert all(p["wrote"] for p in per_rank)                                        # every rank returns after the dataset is on disk
    counts = [len(p["rendered"]) for p in per_rank]
    assert sum(counts) == 12 and max(counts) - min(counts) <= 1                     # 5 + 7 cameras, camera i -> rank i mod N
    assert sorted(x for p in per_rank for x in p["rendered"]) == sorted(RENDERED)   # together: exactly the single-process renders


@pytest.mark.parametrize("mode", ["original", "nosync"])
def test_generate_dataset_sharded_other_modes(tmp_path, standins, mode):
    """world 2: (original) the foreign, distorted dataset cameras still shard camera i -> rank i mod 2 -- r03 serialised them on rank 0;
    (nosync) ranks other than 0 return right after the gather instead of waiting for the serial stage."""
    if mode == "original":
        _original_dataset(tmp_path)   # the photos exist before the ranks start
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    ref, syn = _poses()
    if mode == "original":
        _generator(tmp_path, "single").generate_dataset(_Graph(), ref, original_dataset=_original_dataset(tmp_path))
    else:
        _generator(tmp_path, "single").generate_dataset(_Graph(), ref, synthetic_camera_to_worlds=syn)
    a, b = _tree(tmp_path / "sharded"), _tree(tmp_path / "single")
    a.pop("config.yml"), b.pop("config.yml")
    assert a.keys() == b.keys() and all(a[k] == b[k] for k in a)
    per_rank = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    n = 11 if mode == "original" else 12
    counts = [len(p["rendered"]) for p in per_rank]
    assert sum(counts) == n and max(counts) - min(counts) <= 1
    assert per_rank[0]["wrote"] and (mode == "nosync" or per_rank[1]["wrote"])


def test_rank0_exception_releases_the_idle_ranks(tmp_path, standins):
    """world 2, the diffuser raises on rank 0 inside the serial stage: rank 0 re-raises AFTER meeting rank 1 in the closing barrier, so
    rank 1 returns at once instead of waiting out `serial_stage_timeout_s` (600 s here)."""
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "diffuser_raises"), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in range(2))
    assert r0["raised"] == "diffuser down" and r1["raised"] is None
    assert r1["seconds"] < 60.0


def test_rank0_failure_before_the_serial_stage_releases_the_idle_ranks(tmp_path, standins):
    """world 2, `init_directory` raises on rank 0 (it used to run before the try block that guarantees `_finish`), nothing
    pre-computed: rank 1 is already in the closing barrier (600 s timeout here) and must be released at once."""
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "init_raises"), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in range(2))
    assert r0["raised"] == "cannot create the dataset directory" and r1["raised"] is None
    assert r0["seconds"] < 60.0 and r1["seconds"] < 60.0


@pytest.mark.parametrize("world", [2, 3])
def test_rank0_failure_before_the_precompute_stage_keeps_every_rank_out_of_its_collectives(tmp_path, standins, world):
    """the default (pre-compute ON).  `init_directory` raises on rank 0; the other ranks are about to enter the stage-1
    collectives (the budget all-reduce, then the gathers) and would wait there for rank 0 until the collective timeout.  The failure now
    travels through the budget reduce: nobody renders, rank 0 re-raises its own error, the others raise a RuntimeError, all within seconds."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "init_raises_precompute"), nprocs=world, join=True)
    rs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert rs[0]["raised"] == "cannot create the dataset directory"
    for r in rs[1:]:
        assert r["raised"].startswith("RuntimeError: generate_dataset: rank 0 could not prepare")
    assert all(r["seconds"] < 60.0 and not r["rendered"] for r in rs)
