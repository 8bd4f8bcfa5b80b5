"""Multi-rank path on CPU: world_size 2 and 3 over gloo.  The render itself is replaced by a deterministic
per-camera stand-in (the oracle on a tiny frame for camera 0, a cheap closed form for the rest) -- what is under test is the
sharding, padding, all-gather and re-ordering logic in signerf_amd/sheet.py, which is identical under RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(i, H=6, W=5):
    g = torch.Generator().manual_seed(100 + i)
    rgb = torch.rand(H, W, 3, generator=g)
    depth = torch.full((H, W, 1), float(i)) + torch.rand(H, W, 1, generator=g)
    return rgb, depth


def _worker(rank, world, port, n_cameras, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from signerf_amd import sheet

    rendered = []

    def render_fn(i):
        rendered.append(i)
        return _fake_render(i)

    tiles = sheet.render_cameras_sharded(render_fn, n_cameras)
    torch.save({"tiles": tiles, "rendered": rendered}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_cameras", [(2, 8), (2, 5), (3, 8), (2, 1)])
def test_sharded_sheet_matches_single_rank(tmp_path, world, n_cameras):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_cameras, str(tmp_path)), nprocs=world, join=True)
    want = torch.stack([torch.cat(_fake_render(i), dim=-1) for i in range(n_cameras)])
    seen = []
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert got["tiles"].shape == want.shape
        assert torch.equal(got["tiles"], want), f"rank {r} holds a different sheet"   # every rank has every tile, in camera order
        assert got["rendered"] == list(range(r, n_cameras, world))                    # camera i -> rank i % world
        seen += got["rendered"]
    assert sorted(seen) == list(range(n_cameras))                                     # each camera rendered exactly once


def test_render_views_packs_rgb_mask_condition(monkeypatch):
    """render_views = render_camera per view, packed [rgb | mask | condition]; sharding/gather shared with the sheet path."""
    from signerf_amd import datasetgenerator, sheet

    def fake_render_camera(config, graph, camera, **kw):
        rgb, depth = _fake_render(camera)
        return rgb, depth > float(camera) + 0.5, 1 - depth / 10

    monkeypatch.setattr(datasetgenerator, "render_camera", fake_render_camera)
    t = sheet.render_views(None, [0, 1, 2], None)
    assert t.shape == (3, 6, 5, 5)
    rgb, depth = _fake_render(2)
    assert torch.equal(t[2, ..., :3], rgb) and torch.equal(t[2, ..., 3:4], (depth > 2.5).float()) and torch.equal(t[2, ..., 4:5], 1 - depth / 10)


def test_shard_indices_and_single_process_passthrough():
    from signerf_amd import sheet

    assert sheet.shard_indices(8, 8, 3) == [3]
    assert sheet.shard_indices(9, 8, 0) == [0, 8]
    assert sheet.shard_indices(5, 8, 7) == []
    t = sheet.render_cameras_sharded(lambda i: _fake_render(i), 3)
    assert t.shape == (3, 6, 5, 4) and torch.equal(t[2, ..., :3], _fake_render(2)[0])
