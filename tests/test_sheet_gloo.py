"""Multi-rank path on CPU: world_size 2, 3 and -- r05, the rank count BASELINE configs[2] / [4] name -- 8 over gloo.  The render itself is replaced by a deterministic
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


@pytest.mark.parametrize("world,n_cameras", [(2, 8), (2, 5), (3, 8), (2, 1), (8, 8), (8, 58), (8, 5)])   # 8 x 8: one camera per rank (configs[2]); 58 = 8 + 50 views (configs[4]): 7 x 8 + 2, ragged; 5: ranks without a camera
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


# ---- row-block fallback (one frame over several ranks) ------------------------------------------------------------------
class _FakeBundle:
    def __init__(self, o):
        self.origins = o

    def _map(self, fn):
        return _FakeBundle(fn(self.origins))


class _FakeCamera:
    def __init__(self, H, W):
        self.H, self.W = H, W

    def generate_rays(self, camera_indices=0, aabb_box=None):
        ys, xs = torch.meshgrid(torch.arange(self.H, dtype=torch.float32), torch.arange(self.W, dtype=torch.float32), indexing="ij")
        return _FakeBundle(torch.stack([ys, xs, ys * 0 + 1], dim=-1))


class _FakeModel:
    render_aabb = None

    def __init__(self):
        self.rows_rendered = []

    def get_outputs_for_camera_ray_bundle(self, b):
        o = b.origins  # a pure per-ray function of the pixel coordinates: any block decomposition must reproduce it
        self.rows_rendered.append((int(o[0, 0, 0]), int(o[-1, 0, 0]) + 1))
        return {"rgb": torch.sin(o * 0.37 + 1.0), "depth": (o[..., 0:1] * 1000 + o[..., 1:2])}


def _row_worker(rank, world, port, H, W, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from signerf_amd import sheet

    model = _FakeModel()
    tile = sheet.render_camera_row_sharded(model, _FakeCamera(H, W))
    torch.save({"tile": tile, "rows": model.rows_rendered}, os.path.join(out_dir, f"row_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 40, 12), (3, 50, 7), (2, 7, 5), (3, 8, 4), (8, 100, 6), (8, 24, 4)])
def test_row_sharded_frame_matches_single_rank(tmp_path, world, H, W):
    from signerf_amd import sheet

    port = _free_port()
    mp.spawn(_row_worker, args=(world, port, H, W, str(tmp_path)), nprocs=world, join=True)
    single = sheet.render_camera_row_sharded(_FakeModel(), _FakeCamera(H, W))  # no process group: the plain render
    assert single.shape == (H, W, 4)
    covered = []
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"row_rank{r}.pt"))
        assert torch.equal(got["tile"], single), f"rank {r} holds a different frame"
        covered += got["rows"]
    blocks = [b for b in sheet.row_blocks(H, world) if b[1] > b[0]]
    assert sorted(covered) == blocks and blocks[0][0] == 0 and blocks[-1][1] == H
    assert all(b0[1] == b1[0] for b0, b1 in zip(blocks, blocks[1:]))
    assert all((b[1] - b[0]) % 8 == 0 for b in blocks[:-1])  # whole 8-row tile bands except the last block


def _pipeline_worker(rank, world, port, steps, out_dir):
    """bench.py's depth-1 pipeline: the gather of frame i is in flight while frame i + 1 is 'rendered'."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from signerf_amd import sheet

    done, pending = [], None
    for step in range(steps):
        tile = torch.cat(_fake_render(100 * step + rank), dim=-1)[None]
        handle = sheet.gather_tiles_async(tile, world)
        if pending is not None:
            done.append(pending.wait())
        pending = handle
    done.append(pending.wait())
    torch.save(done, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gathers_deliver_every_frame_in_order(tmp_path):
    world, steps = 2, 4
    mp.spawn(_pipeline_worker, args=(world, _free_port(), steps, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert len(got) == steps
        for step, sheet_tiles in enumerate(got):
            want = torch.stack([torch.cat(_fake_render(100 * step + k), dim=-1) for k in range(world)])
            assert torch.equal(sheet_tiles, want), (r, step)


def test_frame_streams_degrade_to_in_order_execution_on_cpu():
    """sheet.FrameStreams without a GPU (these gloo tests, a CPU-only host): frames run in order on the caller's thread, keep() and join()
    are no-ops."""
    from signerf_amd import sheet

    fs = sheet.FrameStreams(torch.device("cpu"))
    order = []
    for k in range(5):
        with fs.frame(k):
            order.append(k)
            t = fs.keep(torch.full((2,), float(k)))
            assert float(t.sum()) == 2.0 * k
    fs.join()
    assert order == list(range(5))
    fs = sheet.FrameStreams(None)
    with fs.frame(0):
        pass
    fs.join()


# ---- gather strategies: ring all-gather vs direct pushes, identical results -----------------------------------------
def _strategy_worker(rank, world, port, out_dir, n_items=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from signerf_amd import sheet

    n_items = n_items or 2 * world - 1            # ragged: the last rank holds one tile fewer
    mine = sheet.shard_indices(n_items, world, rank)
    local = torch.stack([torch.full((3, 4, 5), float(i)) + torch.arange(5.0) for i in mine])
    got = {}
    for strategy in sheet.GATHER_STRATEGIES:
        for dst in (None, 0, world - 1):
            h = sheet.gather_tiles_async(local, n_items, dst=dst, strategy=strategy)
            t = h.wait()
            got[f"{strategy}/{dst}"] = None if t is None else t.clone()
    # two exchanges in flight at once, waited for out of order (the bench's depth-1 pipeline issues the next before waiting)
    a = sheet.gather_tiles_async(local, n_items, strategy="p2p")
    b = sheet.gather_tiles_async(local * 2, n_items, strategy="p2p")
    got["pipelined"] = torch.stack([a.wait(), b.wait()])
    torch.save(got, os.path.join(out_dir, f"strategy_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 3), (3, 5), (8, 58), (8, 8)])
def test_gather_strategies_deliver_the_same_tiles(tmp_path, world, n_items):
    mp.spawn(_strategy_worker, args=(world, _free_port(), str(tmp_path), n_items), nprocs=world, join=True)
    want = torch.stack([torch.full((3, 4, 5), float(i)) + torch.arange(5.0) for i in range(n_items)])
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"strategy_rank{r}.pt"))
        for key, t in got.items():
            if key == "pipelined":
                assert torch.equal(t[0], want) and torch.equal(t[1], want * 2)
                continue
            strategy, dst = key.split("/")
            if dst == "None" or int(dst) == r:
                assert t is not None and torch.equal(t, want), f"rank {r}: {key}"
            else:
                assert t is None, f"rank {r} is not the root of {key}"


def test_unknown_gather_strategy_raises():
    from signerf_amd import sheet

    with pytest.raises(ValueError, match="one of"):
        sheet.gather_tiles_async(torch.zeros(1, 2, 2, 4), 1, strategy="broadcast")
