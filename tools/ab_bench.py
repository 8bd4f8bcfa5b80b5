#!/usr/bin/env python3
"""Within-process interleaved A/B of render variants selected by environment knobs read per call (e.g. SN_PRIO_MODE).

    python tools/ab_bench.py SN_PRIO_MODE 0 1 2 [--rounds 6] [--config bench|proposal]
Prints median / min kernel ms per variant (HIP events on the launch stream)."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from signerf_amd import Cameras, ops, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("knob")
    ap.add_argument("values", nargs="+")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--config", default="bench")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--implementation", default="torch", help='"tcnn": tiny-cuda-nn grid semantics (weights stay at their random init)')
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = scene.benchmark_config(64) if a.config == "bench" else scene.proposal_config()
    cfg.precision = a.precision
    cfg.implementation = a.implementation
    model = cfg.setup()
    if a.implementation == "torch":
        model.load_state_dict(scene.synthetic_state_dict(cfg), strict=False)
    model = model.to(dev).eval()
    W = H = a.size
    cam = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(dev)[0]
    b = cam.generate_rays(0)
    times = {v: [] for v in a.values}
    for v in a.values:  # warm-up each variant
        os.environ[a.knob] = v
        ops.reload_env(model)   # (the SN_* switches are read at sn_create / sn_finalize_weights, not per render)
        model.get_outputs_for_camera_ray_bundle(b)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v in a.values:
            os.environ[a.knob] = v
            ops.reload_env(model)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.get_outputs_for_camera_ray_bundle(b)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1))
    for v in a.values:
        t = times[v]
        print(f"{a.knob}={v}: median {statistics.median(t):.3f} ms  min {min(t):.3f} ms  ({len(t)} rounds)")


if __name__ == "__main__":
    main()
