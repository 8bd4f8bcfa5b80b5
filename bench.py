#!/usr/bin/env python3
"""bench.py -- ray-samples/s and ms/frame of the reference-sheet render path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = every rank renders ONE 800x800 camera of the 3x3 reference sheet (BASELINE.json configs[1]: synthetic nerfacto
field, hash grid L=16 T=2^19, 64 samples/ray, no proposal nets; rank r renders camera r of circle_poses(8)) through
Cameras.generate_rays -> Model.get_outputs_for_camera_ray_bundle, followed (N>1) by the RCCL all-gather of the finished
[H,W,4] tiles.  Weak scaling: per-GPU work is fixed.  Inputs (weights, camera) are resident in HBM before the timed region.
Prints ONE JSON line (see README / DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_MAIN_SAMPLE = 1024.0  # 16 levels x 8 corners x 2 features x 4 B


def cpu_baseline(cfg, sd, width, height, samples, crop=200):
    """The CPU oracle (a port: nerfstudio's own CPU path cannot be installed) timed on a centred crop of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_config
    from oracle import nerfacto as onf
    from signerf_amd import scene

    c2w = scene.benchmark_cameras(8)[0]
    rays = onf.generate_rays(c2w[:3], float(width), float(width), width / 2, height / 2, height, width)
    y0, x0 = (height - crop) // 2, (width - crop) // 2
    o = rays["origins"][y0:y0 + crop, x0:x0 + crop].contiguous()
    d = rays["directions"][y0:y0 + crop, x0:x0 + crop].contiguous()
    ocfg = oracle_config(cfg)
    onf.get_outputs_for_camera_ray_bundle(sd, ocfg, o[:32], d[:32])  # warm-up
    t = time.perf_counter()
    onf.get_outputs_for_camera_ray_bundle(sd, ocfg, o, d)
    dt = time.perf_counter() - t
    return {"value": crop * crop * samples / dt, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"centred {crop}x{crop} crop of the {width}x{height}x{samples} frame, {dt:.1f} s, torch CPU fp32 oracle",
            "ms_per_frame_extrapolated": dt * 1e3 * (width * height) / (crop * crop)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16x2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from signerf_amd import Cameras, build, scene, sheet

    build.build(verbose=False)
    cfg = scene.benchmark_config(args.samples)
    cfg.precision = args.precision
    sd = scene.synthetic_state_dict(cfg, seed=0)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    W, H, S = args.width, args.height, args.samples
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(dev)
    cam = cams[rank % 8]

    render_ms = []

    def step(timed: bool):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
        e0.record()
        out = model.get_outputs_for_camera_ray_bundle(bundle)
        e1.record()
        tile = torch.cat([out["rgb"], out["depth"]], dim=-1)[None]
        if world > 1:
            tile = sheet.gather_tiles(tile, world)
        if timed:
            render_ms.append((e0, e1))
        return tile

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tiles = step(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in render_ms) / max(len(render_ms), 1)

    if rank == 0:
        samples_per_step = world * W * H * S
        value = samples_per_step * args.steps / elapsed
        achieved = (W * H * S * BYTES_PER_MAIN_SAMPLE) / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "ray-samples/sec (800x800 reference-sheet camera render)",
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32 (fp16 hi+lo split MFMA, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: {W}x{H} rays x {S} samples/ray, nerfacto hash grid L=16 T=2^19 F=2, "
                                   "no proposal nets, random-weight synthetic scene, one camera per GPU + tile all-gather",
                       "rays_per_gpu": W * H, "samples_per_ray": S, "parallelism": f"camera-sharded x{world}"},
            "ms_per_frame": elapsed / args.steps * 1e3,
            "rays_per_sec": world * W * H * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None, "kernel": "sn_render_main_kernel<0>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": W * H * S * BYTES_PER_MAIN_SAMPLE},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, W, H, S)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
