#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: the per-view work of the dataset-generator loops (datasetgenerator.py:331-338, :517-519) --
8 reference cameras (circle_poses) + 50 random_sphere_poses views, each rendered (nerfacto defaults: 256 + 96 proposal + 48 main
samples), masked (aabb mode, 50x50 elliptical dilation) and conditioned -- through signerf_amd.sheet.render_views.

    python tools/views_bench.py [--size 512] [--reps 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from signerf_amd import Cameras, random_sphere_poses, scene, sheet  # noqa: E402
from signerf_amd.datasetgenerator import DatasetGeneratorConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--frames-in-flight", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = scene.proposal_config()
model = cfg.setup()
model.load_state_dict(scene.synthetic_state_dict(cfg, seed=0, density_bias=5.0), strict=False)
model = model.to(dev).eval()
torch.manual_seed(1)
c2w = torch.cat([scene.benchmark_cameras(8), random_sphere_poses(50, torch.device("cpu"), 0.5, (30.0, 120.0), (0.0, 360.0),
                                                               [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])])
S = a.size
cams = Cameras(c2w[:, :3], 1.2 * S, 1.2 * S, S / 2, S / 2, S, S).to(dev)
gen = DatasetGeneratorConfig(aabb_min=[-0.2, -0.2, -0.2], aabb_max=[0.2, 0.2, 0.2])
sheet.render_views(model, cams, gen, frames_in_flight=a.frames_in_flight)
torch.cuda.synchronize()
best = 1e9
for _ in range(a.reps):
    t = time.perf_counter()
    tiles = sheet.render_views(model, cams, gen, frames_in_flight=a.frames_in_flight)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t)
n = len(cams)
print(f"{n} views {S}x{S}, {a.frames_in_flight} in flight (256+96+48 samples, aabb mask 50x50 dilation, condition): {best * 1e3:.1f} ms total, {best * 1e3 / n:.2f} ms per view, "
      f"{n * S * S * 400 / best / 1e9:.1f} G field evaluations/s; mask coverage {float(tiles[..., 3].mean()):.3f}")
