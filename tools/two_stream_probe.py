#!/usr/bin/env python3
"""Does rendering consecutive frames on alternating HIP streams hide the tail of a launch (the last, partly filled round of waves)?
K1 at 800x800 is 10 000 waves on 3072 wave slots = 3.26 rounds; frames are independent, so the head of frame i + 1 can fill the slots the
tail of frame i leaves idle.   python tools/two_stream_probe.py [--workload nerfacto1080]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from signerf_amd import Cameras, scene

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="sheet64"); ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--size", type=int, nargs=2, default=None); a = ap.parse_args()
dev = torch.device("cuda:0")
if a.workload == "sheet64":
    cfg = scene.benchmark_config(64); W, H = a.size or (800, 800); focal = float(W)
else:
    cfg = scene.proposal_config(); W, H = a.size or (1920, 1080); focal = 1.2 * H
model = cfg.setup(); model.load_state_dict(scene.synthetic_state_dict(cfg, seed=0), strict=False); model = model.to(dev).eval()
cam = Cameras(scene.benchmark_cameras(8)[:, :3], focal, focal, W / 2, H / 2, W, H).to(dev)[0]
def run(nstreams, frames, prio=False):
    streams = [torch.cuda.Stream(device=dev, priority=(-1 if (prio and k == 0) else 0)) for k in range(nstreams)]
    outs = []
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(frames):
        with torch.cuda.stream(streams[k % nstreams]):
            out = model.get_outputs_for_camera_ray_bundle(cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb))
            outs.append(out["rgb"][0, 0, 0])
            if len(outs) > 8: outs.pop(0)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / frames * 1e3
def check(nstreams, frames):
    """Every frame of a pipelined run against the one-stream frame, bit for bit (full size: the r01 hazards showed up as a few tiles per frame)."""
    ref = model.get_outputs_for_camera_ray_bundle(cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb))
    ref = {k: ref[k].clone() for k in ("rgb", "depth", "accumulation", "expected_depth")}
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    for st in streams: st.wait_stream(torch.cuda.current_stream())
    kept = []
    for k in range(frames):
        with torch.cuda.stream(streams[k % nstreams]):
            out = model.get_outputs_for_camera_ray_bundle(cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb))
            kept.append({kk: out[kk] for kk in ref})
    torch.cuda.synchronize()
    bad = sum(0 if all(torch.equal(o[kk], ref[kk]) for kk in ref) else 1 for o in kept)
    print(f"{a.workload} {W}x{H}: {frames} frames on {nstreams} streams, frames differing from the one-stream frame: {bad}")
run(1, 10); run(2, 10)
check(2, min(a.frames, 120))
for rep in range(3):
    print(f"{a.workload} {W}x{H}: 1 stream {run(1, a.frames):.4f} ms/frame   2 streams {run(2, a.frames):.4f}   3 streams {run(3, a.frames):.4f}"
          f"   2 streams, one at high priority {run(2, a.frames, True):.4f}")
