#!/usr/bin/env python3
"""What K1 costs in the opt-in single-fp16 mode (precision="fp16", tiny-cuda-nn checkpoints only) and how far its render moves from the
fp32-grade one -- VERDICT r03 item 8.  800x800x64 (BASELINE.json configs[1]'s shape) and 1920x1080 behind the proposal sampler, on a
synthetic tiny-cuda-nn checkpoint (tests/helpers.py::synthetic_tcnn_checkpoint: nerfacto's real grid sizes), one launch at a time.

    python tools/fp16_mode_bench.py [--rounds 30]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import rmse, synthetic_tcnn_checkpoint  # noqa: E402
from signerf_amd import Cameras, scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)


def run(cfg, W, H, focal, tag):
    cfg.implementation = "tcnn"
    cfg.average_init_density = 3.0
    cfg.precision = "fp16"   # (the handle is created with the grid's fp16 storage; the other modes ignore it)
    sd = synthetic_tcnn_checkpoint(cfg, seed=2)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    b = Cameras(scene.benchmark_cameras(8)[:, :3], focal, focal, W / 2, H / 2, W, H).to(dev)[0].generate_rays(0)
    modes = ("fp16x2", "fp16", "fp32")
    out, times = {}, {m: [] for m in modes}
    for m in modes:
        model.config.precision = m
        o = model.get_outputs_for_camera_ray_bundle(b)
        out[m] = {k: o[k].clone() for k in ("rgb", "depth", "accumulation")}
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for m in modes:
            model.config.precision = m
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.get_outputs_for_camera_ray_bundle(b)
            e1.record()
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1))
    med = {m: statistics.median(t) for m, t in times.items()}
    from signerf_amd import ops
    lay = ops.debug_layout(model)
    print(f"    handle: {lay['handle_bytes'] / 1e9:.2f} GB on the device, of which the grid's fp16 storage {lay['half_grid_bytes'] / 1e9:.2f} GB, "
          f"fp32 de-hashed copies {lay['dense_bytes'] / 1e9:.2f} GB")
    print(f"{tag}: render call ms (median of {a.rounds}, interleaved): " + ", ".join(f"{m} {med[m]:.3f}" for m in modes) +
          f"  -> fp16 / fp16x2 = {med['fp16'] / med['fp16x2']:.3f}")
    for ref in ("fp16x2", "fp32"):
        print(f"    fp16 vs {ref}: " + ", ".join(f"{k} rmse {rmse(out['fp16'][k], out[ref][k]):.2e}" for k in ("rgb", "depth", "accumulation")) +
              f"; max |rgb| diff {float((out['fp16']['rgb'] - out[ref]['rgb']).abs().max()):.2e}")
    print(f"    fp16x2 vs fp32 (both fp32-grade): rgb rmse {rmse(out['fp16x2']['rgb'], out['fp32']['rgb']):.2e}; scene: rgb std {float(out['fp32']['rgb'].std()):.3f}, "
          f"accumulation mean {float(out['fp32']['accumulation'].mean()):.3f}")


run(scene.benchmark_config(64), 800, 800, 800.0, "800x800x64, uniform sampler (K1 alone)")
run(scene.proposal_config(), 1920, 1080, 1.2 * 1080, "1920x1080, 256 + 96 + 48 (K2 + K1)")
