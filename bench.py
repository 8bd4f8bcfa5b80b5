#!/usr/bin/env python3
"""bench.py -- ray-samples/s and ms/frame of the reference-sheet render path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = every rank renders ONE 800x800 camera of the 3x3 reference sheet (BASELINE.json configs[1]: synthetic nerfacto
field, hash grid L=16 T=2^19, 64 samples/ray, no proposal nets; rank r renders camera r of circle_poses(8)) through
Cameras.generate_rays -> Model.get_outputs_for_camera_ray_bundle, followed (N>1) by the RCCL all-gather of the finished
[H,W,4] tiles (started asynchronously: it overlaps the next step's render; all K gathers complete inside the timed region).
Weak scaling: per-GPU work is fixed.  Inputs (weights, camera) are resident in HBM before the timed region.
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


def measured_traffic(precision):
    """HBM-side bytes per launch of the dominant kernel, from the committed PMC pass (profiles/traffic.json, written by
    tools/pmc_summary.py --json from TCC_EA0_RDREQ_{32,64,128}B + WRITE_SIZE; bench.py cannot run rocprofv3 on itself)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(precision, {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


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
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": crop * crop * samples / dt, "unit": "ray-samples/s", "cores": torch.get_num_threads(), "kind": "port", "host_cpu": cpu,
            "sample": f"centred {crop}x{crop} crop of the {width}x{height}x{samples} frame, {dt:.1f} s, torch CPU fp32 oracle",
            "ms_per_frame_extrapolated": dt * 1e3 * (width * height) / (crop * crop)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sheet64", choices=["sheet64", "nerfacto1080"],
                    help="sheet64 = BASELINE.json configs[1] (the metric's configuration, default); "
                         "nerfacto1080 = configs[3]: 1920x1080, 2 proposal nets (256 + 96 samples) + 48 main samples")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--precision", default="fp16x2", choices=["fp32", "fp16x2"],
                    help="MFMA arithmetic of the tiny MLPs: fp16x2 = fp32 operands split into fp16 hi+lo, fp32 accumulate "
                         "(measured error identical to exact fp32 MFMA, tests/test_gpu_stages.py); fp32 = exact fp32 MFMA")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the extra (untimed-region) run of the other precision")
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

    # The library ships prebuilt with the snapshot (__graft_entry__.build()).  Only a missing file is rebuilt, by local rank 0
    # alone -- N ranks recompiling into one path at the same time would race.
    if not os.path.exists(build.LIB_PATH):
        if local_rank == 0:
            build.build(verbose=False)
        if world > 1:
            dist.barrier()
    if args.workload == "sheet64":
        args.width, args.height, args.samples = args.width or 800, args.height or 800, args.samples or 64
        cfg = scene.benchmark_config(args.samples)
        focal = float(args.width)
        bytes_per_ray = args.samples * BYTES_PER_MAIN_SAMPLE
    else:
        args.width, args.height = args.width or 1920, args.height or 1080
        cfg = scene.proposal_config()
        args.samples = cfg.num_nerf_samples_per_ray
        focal = 1.2 * args.height
        bytes_per_ray = sum(cfg.num_proposal_samples_per_ray) * 320.0 + args.samples * BYTES_PER_MAIN_SAMPLE  # 161.8 kB
    cfg.precision = args.precision
    sd = scene.synthetic_state_dict(cfg, seed=0)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    W, H, S = args.width, args.height, args.samples
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], focal, focal, W / 2, H / 2, W, H).to(dev)
    cam = cams[rank % 8]

    render_ms = []

    def kernel_ms_of(precision: str, n: int = 5) -> float:
        """Mean HIP-event time of the render call in another arithmetic mode (outside the timed region)."""
        old = model.config.precision
        model.config.precision = precision
        ev = []
        for i in range(n + 1):
            bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            model.get_outputs_for_camera_ray_bundle(bundle)
            b.record()
            if i > 0:
                ev.append((a, b))
        torch.cuda.synchronize()
        model.config.precision = old
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    pending = [None]  # the previous step's tile all-gather, still in flight

    def step(timed: bool):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
        e0.record()
        out = model.get_outputs_for_camera_ray_bundle(bundle)
        e1.record()
        tile = torch.cat([out["rgb"], out["depth"]], dim=-1)[None]
        if timed:
            render_ms.append((e0, e1))
        if world == 1:
            return tile
        # depth-1 pipeline: this frame's all-gather (RCCL's own stream) overlaps the next frame's render; every gather is
        # waited for inside the timed region (drain() below)
        handle = sheet.gather_tiles_async(tile, world)
        done = pending[0].wait() if pending[0] is not None else None
        pending[0] = handle
        return done

    def drain():
        if pending[0] is not None:
            tiles = pending[0].wait()
            pending[0] = None
            return tiles

    model._ensure_engine()  # library load, weight upload and de-hashed copies are set-up, not a step (matters only for --warmup 0)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tiles = step(True)
    tiles = drain() if world > 1 else tiles
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
        achieved = (W * H * bytes_per_ray) / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "ray-samples/sec (%dx%d reference-sheet camera render)" % (W, H),
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else
                     "f32 (MLP operands carried as fp16 hi+lo pairs on the matrix cores, f32 accumulate; measured error equals exact-f32 MFMA)",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[1]: {W}x{H} rays x {S} samples/ray, nerfacto hash grid L=16 T=2^19 F=2, "
                                    "no proposal nets, random-weight synthetic scene, one camera per GPU + tile all-gather")
                       if args.workload == "sheet64" else
                       (f"BASELINE.json configs[3]: {W}x{H} rays, proposal nets 256 + 96 samples (L=5, T=2^17) + {S} main samples "
                        "(L=16, T=2^19), random-weight synthetic scene, one camera per GPU + tile all-gather"),
                       "rays_per_gpu": W * H, "samples_per_ray": S, "parallelism": f"camera-sharded x{world}" + (", tile all-gather overlapped with the next render" if world > 1 else "")},
            "ms_per_frame": elapsed / args.steps * 1e3,
            "rays_per_sec": world * W * H * args.steps / elapsed,
            # SURVEY §8(d) metric (3): every field evaluation of a ray (proposal nets + main field)
            "field_evaluations_per_sec": world * W * H * (S + (sum(cfg.num_proposal_samples_per_ray[:cfg.num_proposal_iterations])
                                                               if args.workload != "sheet64" else 0)) * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": measured_traffic(args.precision) if args.workload == "sheet64" else None,
                         "kernel": ("sn_render_main_kernel<0,%d>" % (0 if args.precision == "fp32" else 1)) if args.workload == "sheet64"
                         else "sn_proposal_kernel + sn_render_main_kernel<1,*> (whole render call)",
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": W * H * bytes_per_ray,
                         "note": "frac > 1 means the algorithmic bytes never reach HBM: the hash table is served by L1/L2/Infinity "
                                 "Cache (`traffic` = measured fabric bytes per launch); the kernel is bound by the L1 gather rate and "
                                 "by SIMD issue (VALU + MFMA serialise on gfx950), see DESIGN.md K1"},
        }
        if args.workload == "sheet64":
            # Secondary view (north_star: "MFMA utilisation against gfx950 peak"): matrix-core instructions issued per launch are
            # fixed by the kernel (per wave-step of 64 samples: 120 v_mfma_f32_32x32x16_f16 in split precision, 320
            # v_mfma_f32_32x32x2_f32 in exact fp32; rocprofv3 SQ_INSTS_MFMA agrees, profiles/) -- issued flops / kernel time
            # against the dense peak of that MFMA type (MI355X_MICROARCH.md: 2.5 PFLOP/s f16, 157.3 TFLOP/s f32-input).
            per_step, flop, peak = (120, 2 * 32 * 32 * 16, 2500.0) if args.precision == "fp16x2" else (320, 2 * 32 * 32 * 2, 157.3)
            issued = (W * H * S / 64) * per_step * flop
            line["roofline_mfma"] = {"bound": "mfma", "achieved": issued / (kernel_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                     "frac": issued / (kernel_ms * 1e-3) / 1e12 / peak,
                                     "algorithmic_tflops": W * H * S * 22784 / (kernel_ms * 1e-3) / 1e12,
                                     "note": "issued MFMA flops (3-term fp16 split, 32-row tile padding) = matrix-pipe busy fraction; "
                                             "algorithmic = 22 784 FLOP per sample (SURVEY 8(d))"}
        if not args.no_alt_precision:
            other = "fp32" if args.precision == "fp16x2" else "fp16x2"
            ms = kernel_ms_of(other)
            line["alt_precision"] = {"precision": other, "kernel_ms": ms, "ray_samples_per_s_per_gpu": W * H * S / (ms * 1e-3),
                                     "roofline_frac": (W * H * bytes_per_ray) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        if world == 1 and not args.no_cpu_baseline and args.workload == "sheet64":
            line["cpu_baseline"] = cpu_baseline(cfg, sd, W, H, S)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
