#!/usr/bin/env python3
"""K1 on a tiny-cuda-nn GRID (implementation="tcnn": the library's positions, dense coarse levels, xor hash) in the fp32-grade default and in
the opt-in single-fp16 mode (precision="fp16": fp16 MLP operands + the grid in fp16 storage -- what tiny-cuda-nn itself computes with;
DESIGN.md "Single-fp16 mode").  Same 800x800x64 frame, synthetic weights of the package's own scene (signerf_amd/scene.py) loaded into the
tcnn-grid model, one launch at a time, interleaved.  Prints ONE JSON object (bench.py puts it under `others`).  No oracle involved: timing only;
how far the mode's render moves is measured by tools/fp16_mode_bench.py and tests/test_gpu_fp16_mode.py.

    python tools/tcnn_modes_bench.py [--rounds 12] [--size 800]
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from signerf_amd import Cameras, ops, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--size", type=int, default=800)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = scene.benchmark_config(64)
    cfg.implementation = "tcnn"
    cfg.precision = "fp16"          # (the handle is created with the grid's fp16 storage; the fp32-grade render ignores it)
    model = cfg.setup()
    model.load_state_dict(scene.synthetic_state_dict(cfg, seed=0), strict=False)
    model = model.to(dev).eval()
    W = H = a.size
    b = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(dev)[0].generate_rays(0)
    modes = ("fp16x2", "fp16")
    out, times = {}, {m: [] for m in modes}
    for m in modes:
        model.config.precision = m
        o = model.get_outputs_for_camera_ray_bundle(b)
        out[m] = {k: o[k].clone() for k in ("rgb", "accumulation")}
        eff = model.effective_precision
        if eff != m:
            raise SystemExit(f"precision {m!r} resolved to {eff!r} for the synthetic weights")
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
    lay = ops.debug_layout(model)
    S = cfg.num_nerf_samples_per_ray
    d = (out["fp16"]["rgb"].double() - out["fp16x2"]["rgb"].double())
    print(json.dumps({
        "workload": f"{W}x{H} rays x {S} samples/ray on a tiny-cuda-nn grid (L=16, T=2^19), synthetic weights, uniform sampler, one launch at a time",
        "fp32_grade_default": {"precision": "fp16x2", "kernel_ms": med["fp16x2"], "ray_samples_per_s": W * H * S / (med["fp16x2"] * 1e-3)},
        "single_fp16_mode": {"precision": "fp16", "kernel_ms": med["fp16"], "ray_samples_per_s": W * H * S / (med["fp16"] * 1e-3),
                             "opt_in": True, "fp32_grade": False,
                             "what": "fp16 MLP operands and activations, the grid read from its fp16 storage (quads + x-pairs: 42 gathers per sample "
                                     "instead of 84); tiny-cuda-nn checkpoints only, never the headline",
                             "rgb_rmse_vs_fp32_grade": float(torch.sqrt(torch.mean(d * d))),
                             "half_grid_bytes": lay["half_grid_bytes"]},
        "ratio": med["fp16"] / med["fp16x2"], "rounds": a.rounds,
        "scene": {"rgb_std": float(out["fp16x2"]["rgb"].std()), "accumulation_mean": float(out["fp16x2"]["accumulation"].mean())}}))


if __name__ == "__main__":
    main()
