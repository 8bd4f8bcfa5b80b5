#!/usr/bin/env python3
"""Renders the same frame N times and reports where repeated renders differ (tile, lane, magnitude).

    python tools/determinism_probe.py [--precision fp16x2] [--n 16] [--config bench|proposal]
A correct kernel prints "all identical"; a missed hardware hazard shows up as a few tiles / a fixed lane range."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from signerf_amd import Cameras, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp16x2")
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--config", default="bench")
    ap.add_argument("--size", type=int, default=800)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = scene.benchmark_config(64) if a.config == "bench" else scene.proposal_config()
    cfg.precision = a.precision
    model = cfg.setup()
    model.load_state_dict(scene.synthetic_state_dict(cfg), strict=False)
    model = model.to(dev).eval()
    W = H = a.size
    cam = Cameras(scene.benchmark_cameras(8)[:, :3], float(W), float(W), W / 2, H / 2, W, H).to(dev)[0]
    b = cam.generate_rays(0)
    ref = None
    bad_runs = 0
    lanes = collections.Counter()
    for i in range(a.n):
        out = model.get_outputs_for_camera_ray_bundle(b)
        img = torch.cat([out["rgb"], out["depth"], out["accumulation"]], dim=-1).clone()
        if ref is None:
            ref = img
            continue
        diff = (img != ref).any(dim=-1)
        n = int(diff.sum())
        if n:
            bad_runs += 1
            ys, xs = torch.nonzero(diff, as_tuple=True)
            mag = float((img - ref).abs().max())
            per = [(int((img[..., c] != ref[..., c]).sum()), float((img[..., c] - ref[..., c]).abs().max())) for c in range(5)]
            print("   per channel r,g,b,depth,acc (count, max):", " ".join(f"{n_}/{m_:.1e}" for n_, m_ in per))
            tiles = {(int(y) // 8, int(x) // 8) for y, x in zip(ys.tolist(), xs.tolist())}
            for y, x in zip(ys.tolist(), xs.tolist()):
                lanes[(y % 8) * 8 + (x % 8)] += 1
            print(f"run {i}: {n} pixels differ in {len(tiles)} tiles, max |diff| {mag:.3e}, first tiles {sorted(tiles)[:4]}")
    if bad_runs == 0:
        print("all identical")
    else:
        print("lanes:", sorted(lanes.items()))


if __name__ == "__main__":
    main()
