#!/usr/bin/env python3
"""Victim = the stage-level kernels (proposal fields) on one stream, aggressor = uniform-sampler renders (the MFMA main kernel)
on another; prints which stage outputs differ from their sequential values (lane histogram).  See DESIGN.md, K2 hazard."""
import os, sys, threading, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_model, small_config
from signerf_amd import ops, Cameras, scene
gpu = torch.device("cuda", 0)
cfg = small_config()
model, _ = make_model(cfg, gpu)
cfg_u = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=48)
model_u = make_model(cfg_u, gpu)[0]
cam = Cameras(scene.benchmark_cameras(8)[:, :3], 280.0, 280.0, 128.0, 96.0, 256, 192).to(gpu)[0]
bundle = cam.generate_rays(0)
g = torch.Generator().manual_seed(0)
R, N, M = 65536, 256, 96
sb = torch.sort(torch.rand(R, N + 1, generator=g), dim=-1).values; sb[:, 0], sb[:, -1] = 0.0, 1.0
w = torch.rand(R, N, generator=g) ** 4
pos = (torch.rand(400000, 3, generator=g) - 0.5) * 3
bins = torch.cumsum(torch.rand(R, 49, generator=g) * 0.1, dim=-1); dens = torch.exp(torch.randn(R, 48, generator=g)); rgbs = torch.rand(R, 48, 3, generator=g)
sb, w, pos, bins, dens, rgbs = [t.to(gpu) for t in (sb, w, pos, bins, dens, rgbs)]
def run_all():
    b, i = ops.pdf_sample(sb, w, M, 0.01); d0, _ = ops.field_forward(model, pos, None, 0); c = ops.composite(bins, dens, rgbs)
    q = (pos * 0.25 + 0.5).clamp(0.001, 0.999)
    d1, _ = ops.field_forward(model, pos, None, 1)
    return {"prop_field0": d0, "prop_field1": d1}
exp = {k: v.clone() for k, v in run_all().items()}; torch.cuda.synchronize()
bad = collections.Counter(); done = threading.Event()
def victim():
    s = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(s):
        for rep in range(60):
            out = run_all(); s.synchronize()
            for k in exp:
                n = int((out[k] != exp[k]).sum())
                if n:
                    bad[k] += n
                    if bad[k] == n:
                        idx = torch.nonzero((out[k] != exp[k]).reshape(out[k].shape[0], -1).any(dim=1)).flatten()
                        print(k, "first differing rows: lanes", sorted(collections.Counter((idx % 64).tolist()).items())[:20], flush=True)
def aggressor():
    s = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(s):
        while not done.is_set():
            for _ in range(4): model_u.get_outputs_for_camera_ray_bundle(bundle)
            s.synchronize()
ta, tv = threading.Thread(target=aggressor), threading.Thread(target=victim)
ta.start(); tv.start(); tv.join(); done.set(); ta.join()
print("differences:", dict(bad) or "none")
