"""BASELINE.json configs[4] with REAL renders: ``DatasetGenerator.generate_dataset``
(/root/reference/signerf/datasetgenerator/datasetgenerator.py:185-393) -- 8 reference cameras (3x3 sheet) + synthetic views through the
HIP render / mask / condition / resize / uint8 kernels.  (1) Two processes (own HIP contexts and handles, both on the box's one GPU,
process group gloo: tiles staged through the host) write the SAME bytes as one process; (2) what lands on disk is what the stage
functions produce (pre-computed views == views rendered inside the loop)."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SIZE, N_VIEWS = 96, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(dev):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_model, small_config
    from signerf_amd import random_sphere_poses, scene

    cfg = small_config(num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=24)
    model, _ = make_model(cfg, dev, density_bias=5.0)
    ref = scene.benchmark_cameras(8)[:, :3]
    torch.manual_seed(1)
    syn = random_sphere_poses(N_VIEWS, torch.device("cpu"), 0.5, (30.0, 120.0), (0.0, 360.0), [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])[:, :3]
    return model, ref, syn


def _generator(path, name, dev, **kw):
    from signerf_amd.datasetgenerator import DatasetGenerator, DatasetGeneratorConfig

    cfg = DatasetGeneratorConfig(path=path, dataset_name=name, fx=1.2 * SIZE, fy=1.2 * SIZE, cx=SIZE / 2, cy=SIZE / 2, width=SIZE, height=SIZE,
                                 rows=3, cols=3, mask_dialation=(7, 7))   # the default +-0.1 box
    return DatasetGenerator(cfg, torch.eye(4)[:3], 1.0, None, device=dev, **kw)


def _tree(root):
    out = {}
    for d, _, files in os.walk(root):
        for f in files:
            p = os.path.join(d, f)
            out[os.path.relpath(p, root)] = open(p, "rb").read()
    out.pop("config.yml", None)   # (holds the dataset name)
    return out


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, ref, syn = _setup(dev)
    n = []
    orig = model.get_outputs_for_camera_ray_bundle
    model.get_outputs_for_camera_ray_bundle = lambda b: (n.append(1), orig(b))[1]
    _generator(out_dir, "sharded", dev).generate_dataset(model, ref, synthetic_camera_to_worlds=syn)
    json.dump({"renders": len(n)}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.destroy_process_group()


def test_two_processes_write_the_single_process_dataset(gpu, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    model, ref, syn = _setup(gpu)
    _generator(tmp_path, "single", gpu).generate_dataset(model, ref, synthetic_camera_to_worlds=syn)
    _generator(tmp_path, "inloop", gpu, precompute=False, save_workers=0).generate_dataset(model, ref, synthetic_camera_to_worlds=syn)
    a, b, c = _tree(tmp_path / "sharded"), _tree(tmp_path / "single"), _tree(tmp_path / "inloop")
    assert a.keys() == b.keys() == c.keys() and len(a) == 1 + 4 + 8 * (8 + N_VIEWS)
    for k in a:
        assert a[k] == b[k], f"{k}: two-process dataset differs from the single-process one"
        assert c[k] == b[k], f"{k}: renders inside the serial loop differ from the pre-computed ones"
    assert [json.load(open(tmp_path / f"rank{r}.json"))["renders"] for r in range(2)] == [7, 7]   # 8 + 6 cameras, i -> rank i % 2
    t = json.load(open(tmp_path / "single" / "transforms.json"))
    assert t["reference_indices"] == list(range(8)) and t["generated_indices"] == list(range(8, 8 + N_VIEWS))


def test_files_hold_what_the_stage_functions_produce(gpu, tmp_path):
    """images/, rendered/, masks/, conditions/ of a generated view against render_camera + the sheet functions called directly; the
    mask is non-trivial with the DEFAULT +-0.1 box (neither empty nor the whole frame)."""
    from PIL import Image

    from signerf_amd import Cameras
    from signerf_amd.dataset_io import tensor_to_uint8
    from signerf_amd.datasetgenerator import render_camera

    model, ref, syn = _setup(gpu)
    gen = _generator(tmp_path, "exp", gpu)
    gen.generate_dataset(model, ref, synthetic_camera_to_worlds=syn)
    root = tmp_path / "exp"
    cams = Cameras(syn, 1.2 * SIZE, 1.2 * SIZE, SIZE / 2, SIZE / 2, SIZE, SIZE).to(gpu)
    cover = []
    for k in (0, N_VIEWS - 1):
        rgb, mask, cond = render_camera(gen.config, model, cams[k])
        idx = 8 + k
        assert np.array_equal(np.array(Image.open(root / "rendered" / f"image_{idx}.png")), tensor_to_uint8(rgb).cpu().numpy())
        assert np.array_equal(np.array(Image.open(root / "masks" / f"mask_{idx}.png")), tensor_to_uint8(mask.float()).cpu().numpy()[..., 0])
        assert np.array_equal(np.array(Image.open(root / "conditions" / f"condition_{idx}.png")), tensor_to_uint8(cond).cpu().numpy()[..., 0])
        cover.append(float(mask.float().mean()))
        # identity diffuser: the edited image is the up-scaled down-scaled render (:653-659)
        from signerf_amd.ops import resize_bilinear

        up = resize_bilinear(resize_bilinear(rgb, SIZE // 2, SIZE // 2), SIZE, SIZE)
        assert np.array_equal(np.array(Image.open(root / "images" / f"image_{idx}.png")), tensor_to_uint8(up).cpu().numpy())
    assert all(0.02 < c < 0.9 for c in cover), cover
    sheet = np.array(Image.open(root / "references" / "image_reference_sheet.png"))
    assert sheet.shape == (3 * SIZE // 2, 3 * SIZE // 2, 3) and (sheet[SIZE:, SIZE:] == 255).all()   # empty last cell of the first sheet
