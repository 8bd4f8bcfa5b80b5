#!/usr/bin/env python3
"""The 1920x1080 nerfacto frame (K2 + K1 behind it) on operands that do not toggle -- the K1 experiment of tools/power_ab.py (`zero_both`) for the
proposal kernel: the same binary and instruction stream (early termination off: nothing may be skipped), MLP matrices 0 (biases kept) and every hash
table one constant.  Run under rocprofv3 --kernel-trace --stats for the per-kernel times:

    SN_EARLY_TERM=0 rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/data_activity_frame.py --scene base|constant
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # docs/history/tools/ -> repository root
sys.path.insert(0, ROOT)
from signerf_amd import Cameras, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="base", choices=["base", "constant"])
    ap.add_argument("--seconds", type=float, default=5.0)
    a = ap.parse_args()
    assert os.environ.get("SN_EARLY_TERM") == "0", "run with SN_EARLY_TERM=0: a constant medium would otherwise terminate early and skip work"
    dev = torch.device("cuda", 0)
    cfg = scene.proposal_config()
    sd = scene.synthetic_state_dict(cfg, seed=0)
    if a.scene == "constant":
        for k in sd:
            if k.endswith(".weight") and ".layers." in k:
                sd[k] = torch.zeros_like(sd[k])
            if k.endswith("hash_table"):
                sd[k] = torch.full_like(sd[k], 0.5)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    W, H = 1920, 1080
    cam = Cameras(scene.benchmark_cameras(8)[:, :3], 1.2 * H, 1.2 * H, W / 2, H / 2, W, H).to(dev)[0]
    b = cam.generate_rays(0)
    for _ in range(3):
        model.get_outputs_for_camera_ray_bundle(b)
    torch.cuda.synchronize()
    ev, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.get_outputs_for_camera_ray_bundle(b)
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev[len(ev) // 2:])
    print(f"scene {a.scene}: {len(ev)} frames, median of the second half {ms[len(ms) // 2]:.3f} ms per launch (K2 + K1), effective precision {model.effective_precision}")


if __name__ == "__main__":
    main()
