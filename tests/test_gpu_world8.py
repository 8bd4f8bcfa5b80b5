"""World-2 / -4 / -8 rehearsal on ONE GPU (r06: the four launch shapes of the driver's 1 / 2 / 4 / 8 curve have all executed): BASELINE.json configs[2] (3x3 sheet: 8 cameras on 8 MI355X, RCCL tile all-gather) and
the 8-GPU leg of configs[4] (DatasetGenerator's 8 + 50 views, /root/reference/signerf/datasetgenerator/datasetgenerator.py:331,517-519) at
the RANK COUNT they name.  The GPU box has one GPU, so the eight ranks share cuda:0 and the process group is gloo (tiles staged through
the host; RCCL refuses several ranks per device): ownership (camera i -> rank i mod 8, `per = 1`), the ragged 58 = 7 x 8 + 2 split, every
gather strategy, gather-to-root, re-ordering and the all-ranks-exit protocol are the code that runs over RCCL.  Small frames and
`dense_levels = -1` (no de-hashed copies) so that eight handles are cheap.  No scaling number is claimed from this: eight processes on one
GPU measure nothing about xGMI."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
WORLD, SIZE, N_VIEWS = 8, 64, 50


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev, proposals=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_model, small_config

    kw = dict(num_proposal_samples_per_ray=(48, 24), num_nerf_samples_per_ray=16) if proposals else dict(num_proposal_iterations=0, num_nerf_samples_per_ray=24)
    cfg = small_config(dense_levels=-1, **kw)
    model, _ = make_model(cfg, dev, density_bias=5.0)
    return model.eval()


# ---- configs[2]: the 8-camera sheet, one camera per rank, every exchange ----------------------------------------------------------------
def _sheet_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model(dev, proposals=False)
    from signerf_amd import Cameras, scene, sheet

    cams = Cameras(scene.benchmark_cameras(8)[:, :3], float(SIZE), float(SIZE), SIZE / 2, SIZE / 2, SIZE, SIZE).to(dev)
    rendered = []

    def render_fn(i):
        rendered.append(i)
        out = model.get_outputs_for_camera_ray_bundle(cams[i].generate_rays(camera_indices=0, aabb_box=model.render_aabb))
        return out["rgb"], out["depth"]

    got = {}
    for strategy in sheet.GATHER_STRATEGIES:
        for dst in (None, 0):
            t = sheet.render_cameras_sharded(render_fn, 8, device=dev, dst=dst, strategy=strategy)
            got[f"{strategy}/{dst}"] = None if t is None else t.cpu()
    got["row_sharded"] = sheet.render_camera_row_sharded(model, cams[3]).cpu()        # one frame over 8 ranks: 8 rows each
    got["n_renders"] = len(rendered)
    got["owned"] = sorted(set(rendered))
    local = []
    for i in range(8):   # what ONE process renders, in this process
        out = model.get_outputs_for_camera_ray_bundle(cams[i].generate_rays(camera_indices=0, aabb_box=model.render_aabb))
        local.append(torch.cat([out["rgb"], out["depth"]], dim=-1))
    got["local"] = torch.stack(local).cpu()
    torch.cuda.synchronize()
    torch.save(got, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"rank{rank}.done"), "w").write("ok")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_world8_reference_sheet_every_strategy(gpu, tmp_path, world):
    """The 8-camera sheet over 2, 4 and 8 ranks (4, 2, 1 cameras per rank): the launch shapes of the driver's scaling curve."""
    from signerf_amd import sheet

    mp.spawn(_sheet_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    single = got[0]["local"]
    assert single.shape == (8, SIZE, SIZE, 4) and float(single[..., :3].std()) > 0.05
    for r in range(world):
        assert os.path.exists(os.path.join(tmp_path, f"rank{r}.done")), f"rank {r} did not exit cleanly"
        # camera i -> rank i mod world, each once per exchange
        assert got[r]["owned"] == list(range(r, 8, world)) and got[r]["n_renders"] == (8 // world) * 2 * len(sheet.GATHER_STRATEGIES)
        assert torch.equal(got[r]["local"], single)
        for strategy in sheet.GATHER_STRATEGIES:
            assert torch.equal(got[r][f"{strategy}/None"], single), f"rank {r}: {strategy} all-gather differs from the single-process sheet"
            t = got[r][f"{strategy}/0"]
            assert (t is None) if r else torch.equal(t, single), f"rank {r}: {strategy} gather-to-root"
        assert torch.equal(got[r]["row_sharded"], single[3])


# ---- configs[4]: DatasetGenerator.generate_dataset, 8 reference + 50 views, world 8 --------------------------------------------------
def _gen_setup(dev):
    model = _model(dev)
    from signerf_amd import random_sphere_poses, scene

    ref = scene.benchmark_cameras(8)[:, :3]
    torch.manual_seed(1)
    syn = random_sphere_poses(N_VIEWS, torch.device("cpu"), 0.5, (30.0, 120.0), (0.0, 360.0), [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])[:, :3]
    return model, ref, syn


def _generator(path, name, dev, **kw):
    from signerf_amd.datasetgenerator import DatasetGenerator, DatasetGeneratorConfig

    cfg = DatasetGeneratorConfig(path=path, dataset_name=name, fx=1.2 * SIZE, fy=1.2 * SIZE, cx=SIZE / 2, cy=SIZE / 2, width=SIZE, height=SIZE,
                                 rows=3, cols=3, mask_dialation=(5, 5))
    return DatasetGenerator(cfg, torch.eye(4)[:3], 1.0, None, device=dev, **kw)


def _gen_worker(rank, world, port, out_dir, strategy):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SIGNERF_GATHER"] = strategy
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, ref, syn = _gen_setup(dev)
    n = []
    orig = model.get_outputs_for_camera_ray_bundle
    model.get_outputs_for_camera_ray_bundle = lambda b: (n.append(1), orig(b))[1]
    _generator(out_dir, f"world8_{strategy}", dev).generate_dataset(model, ref, synthetic_camera_to_worlds=syn)
    json.dump({"renders": len(n)}, open(os.path.join(out_dir, f"{strategy}_rank{rank}.json"), "w"))
    dist.destroy_process_group()


def _tree(root):
    out = {}
    for d, _, files in os.walk(root):
        for f in files:
            p = os.path.join(d, f)
            out[os.path.relpath(p, root)] = open(p, "rb").read()
    out.pop("config.yml", None)
    return out


@pytest.mark.parametrize("strategy", ["all_gather", "p2p", pytest.param("all_to_all", marks=pytest.mark.slow)])
def test_world8_generate_dataset_58_views(gpu, tmp_path, strategy):
    """8 + 50 cameras over 8 ranks (58 = 7 x 8 + 2: ranks 0 and 1 own eight views, the others seven), tiles gathered to rank 0, which alone
    runs the serial diffusion sequence and writes the files: byte-identical to the single-process dataset, every rank exits."""
    mp.spawn(_gen_worker, args=(WORLD, _free_port(), str(tmp_path), strategy), nprocs=WORLD, join=True)
    model, ref, syn = _gen_setup(gpu)
    _generator(tmp_path, "single", gpu).generate_dataset(model, ref, synthetic_camera_to_worlds=syn)
    a, b = _tree(tmp_path / f"world8_{strategy}"), _tree(tmp_path / "single")
    assert a.keys() == b.keys() and len(a) == 1 + 4 + 8 * (8 + N_VIEWS)
    for k in a:
        assert a[k] == b[k], f"{k}: the world-8 dataset ({strategy}) differs from the single-process one"
    renders = [json.load(open(tmp_path / f"{strategy}_rank{r}.json"))["renders"] for r in range(WORLD)]
    assert renders == [8, 8, 7, 7, 7, 7, 7, 7], renders


# ---- bench.py at the driver's N = 8 command line (gloo, ranks share the GPU) ------------------------------------------------------------------
@pytest.mark.parametrize("world,scaling", [(8, "weak"), (8, "strong"), (4, "weak"), (2, "weak"), (2, "strong")])
def test_bench_world8_dry_run(gpu, world, scaling):
    """bench.py at the driver's N = 2 / 4 / 8 command lines (gloo, the ranks share the GPU): the line, the r06 preflight (every rank
    identified over the process group before the first render) and the post-run checks (per-rank one-launch times, the gathered tiles bit
    for bit against local renders)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--width", "96", "--height", "96", "--no-cpu-baseline", "--no-alt-precision", "--scaling", scaling]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == scaling and d["config"]["dist_world_size"] == world and d["config"]["ranks_share_a_gpu"] is True
    cams_per_step = 8 if scaling == "strong" else world
    assert d["config"]["cameras_per_step"] == cams_per_step
    assert abs(d["value"] - cams_per_step * 96 * 96 * 64 * 3 / d["timed_region_s"]) / d["value"] < 1e-6
    c = d["config"]
    assert c["preflight"] == "ok" and c["distinct_devices"] == 1 and c["devices"].count("r") >= world     # (one shared GPU here; RCCL asserts N distinct)
    assert c["gathered_tiles_bit_identical"] is True and d["gathered_tiles_check"]["tiles_checked_per_rank"] == world
    assert 0 < c["rank_kernel_ms_min"] <= c["rank_kernel_ms_mean"] <= c["rank_kernel_ms_max"] and 0 <= c["slowest_rank"] < world
    assert len(d["rank_kernel_ms"]["per_rank"]) == world and all(v > 0 for v in d["rank_kernel_ms"]["per_rank"])
    assert set(d["gather_ms"]["exposed_by_strategy"]) == {"all_gather", "p2p", "all_to_all"}
    assert all(v is not None and v > 0 for v in d["gather_ms"]["exposed_by_strategy"].values())
