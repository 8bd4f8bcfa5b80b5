#!/usr/bin/env python3
"""Where does a SHORT timed window of bench.py lose time?  The driver times `bench.py --gpus 1 --steps 20 --warmup 5` (54 ms of render); the
builder's default is 300 steps.  Same loop as bench.py's `step()` (two frames in flight on alternating streams), K = 10 .. 160: elapsed =
a + b K separates the per-window overhead a from the frame period b, and the HIP-event timeline of one K = 20 window shows where a sits.

    python tools/short_window_probe.py
"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # docs/history/tools/ -> repository root
sys.path.insert(0, ROOT)
from signerf_amd import Cameras, scene, sheet  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = scene.benchmark_config(64)
    model = cfg.setup()
    model.load_state_dict(scene.synthetic_state_dict(cfg, seed=0), strict=False)
    model = model.to(dev).eval()
    cam = Cameras(scene.benchmark_cameras(8)[:, :3], 800.0, 800.0, 400.0, 400.0, 800, 800).to(dev)[0]
    frames = sheet.FrameStreams(dev, 2)
    issued = [0]

    def step(record=None):
        with frames.frame(issued[0]):
            bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
            if record is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            out = model.get_outputs_for_camera_ray_bundle(bundle)
            if record is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                record.append((e0, e1))
            tile = frames.keep(torch.cat([out["rgb"], out["depth"]], dim=-1)[None])
        issued[0] += 1
        return tile

    def window(K, warmup=5, timeline=False):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        rec = [] if timeline else None
        start = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if timeline:
            start.record()
        host = []
        for _ in range(K):
            step(rec)
            host.append(time.perf_counter() - t0)
        frames.join()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if timeline:
            rows = [(start.elapsed_time(a), start.elapsed_time(b)) for a, b in rec]
            return el, t_issue, host, rows
        return el, t_issue

    for _ in range(3):
        window(20)
    print("K    elapsed ms   per step   host issue ms")
    pts = []
    for K in (10, 20, 40, 80, 160, 20, 10, 40):
        el = [window(K) for _ in range(5)]
        e = statistics.median(x[0] for x in el) * 1e3
        i = statistics.median(x[1] for x in el) * 1e3
        pts.append((K, e))
        print(f"{K:4d} {e:10.3f} {e / K:10.4f} {i:10.3f}")
    # least squares a + b K
    n = len(pts)
    sx, sy = sum(k for k, _ in pts), sum(e for _, e in pts)
    sxx, sxy = sum(k * k for k, _ in pts), sum(k * e for k, e in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    a = (sy - b * sx) / n
    print(f"fit: elapsed = {a:.3f} ms + {b:.4f} ms x K")
    el, t_issue, host, rows = window(20, timeline=True)
    print(f"timeline of one K = 20 window (elapsed {el * 1e3:.3f} ms, all launches issued after {t_issue * 1e3:.3f} ms of host time):")
    for k, ((a0, a1), h) in enumerate(zip(rows, host)):
        print(f"  frame {k:2d}: host returned at {h * 1e3:7.3f} ms | GPU start {a0:7.3f}  end {a1:7.3f}  ({a1 - a0:.3f} ms)")


if __name__ == "__main__":
    main()
