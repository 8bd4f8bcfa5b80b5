#!/usr/bin/env python3
"""Value per byte of the de-hashed copies of the main grid: K1 time over the 8 reference-sheet cameras and the
HBM footprint of the copies for 0 / 8 / 9 / 10 / 11 copied levels, interleaved in one process.

    python tools/dense_sweep.py [--reps 3] [--frames 6]
"""
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
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--frames", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = scene.benchmark_config(64)
    model = cfg.setup()
    model.load_state_dict(scene.synthetic_state_dict(cfg), strict=False)
    model = model.to(dev).eval()
    W = H = 800
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], 800.0, 800.0, 400.0, 400.0, W, H).to(dev)
    bundles = [cams[i].generate_rays(0) for i in range(8)]
    configs = [(lv, 1) for lv in (0, 8, 9, 10, 11)]   # (orientation sets, the second column of r02's sweep, were removed afterwards)
    times = {c: [[] for _ in range(8)] for c in configs}
    bytes_ = {}
    for rep in range(a.reps):
        for lv, st in configs:
            os.environ["SN_DENSE_LEVELS"] = str(lv)
            model.mark_weights_dirty()
            model.get_outputs_for_camera_ray_bundle(bundles[0])  # re-finalize + warm
            bytes_[(lv, st)] = ops.debug_layout(model, -1)["dense_bytes"]
            for cam in range(8):
                model.get_outputs_for_camera_ray_bundle(bundles[cam])
                ev = []
                for _ in range(a.frames):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    model.get_outputs_for_camera_ray_bundle(bundles[cam])
                    e1.record()
                    ev.append((e0, e1))
                torch.cuda.synchronize()
                times[(lv, st)][cam] += [x.elapsed_time(y) for x, y in ev]
    best = min(statistics.mean(statistics.median(t) for t in times[c]) for c in configs)
    print("levels sets  copies[MB]   mean-of-8-cameras[ms]  vs best   camera 0 [ms]  slowest camera [ms]   per camera")
    for c in configs:
        per_cam = [statistics.median(t) for t in times[c]]
        m = statistics.mean(per_cam)
        print(f"{c[0]:6d} {c[1]:4d}  {bytes_[c] / 1e6:10.1f}   {m:21.3f}  {100 * (m / best - 1):+6.1f}%   {per_cam[0]:13.3f}  {max(per_cam):19.3f}   "
              + " ".join(f"{x:.2f}" for x in per_cam))


if __name__ == "__main__":
    main()
