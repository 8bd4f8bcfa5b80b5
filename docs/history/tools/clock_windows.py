#!/usr/bin/env python3
"""How the clock reading of `sn_clock_probe` depends on the window it is taken over (r03): 12 / 60 / 200 / 330 frames of the 800x800x64 bench
render, with the GPU idle before or 300 frames already queued.  The readings agree within ~5 % (1.75-1.91 GHz on the r03 box); a probe
whose wave only runs once the render queue has drained reads the idle boost clock (2.4 GHz) -- `tools/power_ab.py` therefore synchronises
the render STREAM, not the device, while its probe is in flight.

    python tools/clock_windows.py       (GPU box)
"""
import sys, os, statistics, time
sys.path.insert(0, os.getcwd())
import torch
from signerf_amd import Cameras, _lib, scene
dev = torch.device("cuda", 0)
cfg = scene.benchmark_config(64)
model = cfg.setup(); model.load_state_dict(scene.synthetic_state_dict(cfg, seed=0), strict=False); model = model.to(dev).eval()
cam = Cameras(scene.benchmark_cameras(8)[:, :3], 800.0, 800.0, 400.0, 400.0, 800, 800).to(dev)[0]
b = cam.generate_rays(0)
lib = _lib.load()
for _ in range(20): model.get_outputs_for_camera_ray_bundle(b)
torch.cuda.synchronize()
def probe(frames, pre=0, secs=None):
    out = torch.zeros(3, dtype=torch.int64, device=dev); side = torch.cuda.Stream(device=dev)
    for _ in range(pre): model.get_outputs_for_camera_ray_bundle(b)
    if pre == 0: torch.cuda.synchronize()
    _lib.check(lib.sn_clock_probe(out.data_ptr(), secs or min(0.9, frames*2.85e-3*0.8), side.cuda_stream), None, "p")
    ev=[]
    for _ in range(frames):
        a,c=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); model.get_outputs_for_camera_ray_bundle(b); c.record(); ev.append((a,c))
    torch.cuda.synchronize()
    cyc,ticks,rate=(int(x) for x in out.tolist())
    return cyc/(ticks/rate)/1e9, statistics.median(x.elapsed_time(y) for x,y in ev)
for frames, pre in ((12,0),(12,0),(60,0),(200,0),(330,0),(12,300),(60,300),(330,300)):
    g,ms = probe(frames, pre)
    print(f"probe over {frames:4d} frames ({'after %d queued frames' % pre if pre else 'GPU idle before'}): {g:.3f} GHz, median launch {ms:.3f} ms")
    time.sleep(1.0)
