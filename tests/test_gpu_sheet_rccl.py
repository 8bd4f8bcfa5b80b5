"""The RCCL leg of signerf_amd/sheet.py on the one GPU a test box has: a single-rank "nccl" (= RCCL on ROCm) process group runs the
same all_gather_into_tensor / async-work code path as the N-GPU bench (the multi-rank logic itself is covered on CPU over gloo,
tests/test_sheet_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from signerf_amd import sheet

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_overlapped_tile_gather_over_rccl(gpu):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        g = torch.Generator(device="cpu").manual_seed(0)
        frames = [torch.rand(1, 64, 48, 4, generator=g).to(gpu) for _ in range(4)]
        done, pending = [], None
        side = torch.zeros(1024, 1024, device=gpu)
        for f in frames:                       # bench.py's depth-1 pipeline
            h = sheet.gather_tiles_async(f, 1)
            side = side @ side                 # "the next render": work on the caller's stream while the gather is in flight
            if pending is not None:
                done.append(pending.wait())
            pending = h
        done.append(pending.wait())
        torch.cuda.synchronize()
        assert len(done) == 4 and all(torch.equal(a, b) for a, b in zip(done, frames))
        assert torch.equal(sheet.gather_tiles(frames[0], 1), frames[0])
    finally:
        dist.destroy_process_group()


def test_bench_preflight_and_rank_spread_over_rccl(gpu):
    """bench.py's N > 1 preflight (r06) on the one backend it is written for: a single-rank RCCL group runs its all_gather_object, the RCCL
    version query, the device identity and the device-side all-reduce -- the calls the first real 8-GPU run makes before its first render."""
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        pre = bench.preflight(1, 0, gpu, "nccl", False)
        assert pre["preflight"] == "ok" and pre["distinct_devices"] == 1 and pre["devices"].startswith("r0 ")
        assert pre["rccl_version"] and pre["rccl_version"][0].isdigit(), pre      # RCCL reports a version through torch.cuda.nccl
        assert pre["device_cus"] == 256
        ident = bench.device_identity(gpu)
        assert ident["pci"].count(":") == 2 and ident["cus"] == 256
        sp = bench.rank_spread(2.75, 1, gpu, "nccl")
        assert sp == {"min": 2.75, "max": 2.75, "mean": 2.75, "slowest_rank": 0, "per_rank": [2.75]}
    finally:
        dist.destroy_process_group()
