#!/usr/bin/env python3
"""Two host threads on two streams render through ONE model (the viewer + generator situation, SURVEY §8(b) "Threading") and
every frame is compared with its sequential render.  Prints where frames differ (key, pixel count, 8x8-tile lane histogram)."""
import argparse, collections, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_model, small_config
from signerf_amd import Cameras, scene

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=40); ap.add_argument("--props", type=int, default=2)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--mode", default="shared", help="shared | two-models | serial (host lock around render+sync)"); a = ap.parse_args()
gpu = torch.device("cuda", 0)
cfg = small_config(num_proposal_iterations=a.props, num_proposal_samples_per_ray=((48, 24) if a.props == 2 else (48,) if a.props == 1 else ()),
                   num_nerf_samples_per_ray=16)
model, _ = make_model(cfg, gpu)
model2 = make_model(cfg, gpu)[0] if a.mode == 'two-models' else model
if a.mode == 'vs-uniform':  # thread 1 renders with the uniform sampler (main kernel only), thread 0 with the proposal path
    cfg_u = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=48)
    model2 = make_model(cfg_u, gpu)[0]
glock = threading.Lock()
W, H = a.size, (a.size * 3) // 4
cams = Cameras(scene.benchmark_cameras(8)[:, :3], 1.1 * W, 1.1 * W, W / 2, H / 2, W, H).to(gpu)
bundles = [cams[i].generate_rays(0) for i in range(4)]
expect = [{k: v.clone() for k, v in model.get_outputs_for_camera_ray_bundle(b).items()} for b in bundles]
expect2 = expect if a.mode != 'vs-uniform' else [{k: v.clone() for k, v in model2.get_outputs_for_camera_ray_bundle(b).items()} for b in bundles]
torch.cuda.synchronize()
shown = []
bad = collections.Counter(); lanes = collections.Counter(); lock = threading.Lock()
def worker(tid):
    s = torch.cuda.Stream(device=gpu)
    if a.mode == 'other-work' and tid == 1:
        with torch.cuda.stream(s):
            n = int(os.environ.get('SN_PROBE_MM', '4096'))
            dt = torch.float16 if os.environ.get('SN_PROBE_MM_HALF') else torch.float32
            x = torch.rand(64, n, n, device=gpu, dtype=dt) if n <= 512 else torch.rand(n, n, device=gpu, dtype=dt)
            while not done.is_set():
                for _ in range(20):
                    x = (x @ x).clamp_(0, 1) * 0.5 + 0.1
                s.synchronize()
        return
    with torch.cuda.stream(s):
        for rep in range(a.reps):
            for i in (range(4) if tid == 0 else reversed(range(4))):
                m = model if tid == 0 else model2
                if a.mode == 'serial':
                    with glock:
                        out = m.get_outputs_for_camera_ray_bundle(bundles[i]); s.synchronize()
                else:
                    out = m.get_outputs_for_camera_ray_bundle(bundles[i])
                    s.synchronize()
                for k in ("prop_depth_0", "prop_depth_1", "expected_depth", "rgb", "depth", "accumulation"):
                    if k not in out: continue
                    ex = expect if tid == 0 else expect2
                    d = (out[k] != ex[i][k]).any(dim=-1)
                    n = int(d.sum())
                    if n:
                        with lock:
                            bad[(tid, k)] += n
                            ys, xs = torch.nonzero(d, as_tuple=True)
                            for y, x in zip(ys.tolist(), xs.tolist()): lanes[(y % 8) * 8 + x % 8] += 1
                            if k == 'rgb' and len(shown) < 3:
                                shown.append(1)
                                tiles = sorted({(y // 8, x // 8) for y, x in zip(ys.tolist(), xs.tolist())})
                                print('   tiles (ty,tx):', tiles[:40], 'n_tiles', len(tiles), 'of', ((H + 7) // 8) * ((W + 7) // 8), flush=True)
                            if sum(bad.values()) < 2000: print(f"thread {tid} rep {rep} cam {i} {k}: {n} pixels differ, max {float((out[k]-ex[i][k]).abs().max()):.3e}", flush=True)
done = threading.Event()
ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
[t.start() for t in ts]; ts[0].join(); done.set(); ts[1].join()
print("differences:", dict(bad) or "none"); 
if lanes: print("lanes:", sorted(lanes.items()))
