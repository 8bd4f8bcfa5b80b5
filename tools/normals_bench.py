#!/usr/bin/env python3
"""Times the normals kernel (row a16, csrc/sn_normals.h) next to the colour render on the bench workloads.

    python tools/normals_bench.py [--workload sheet64|nerfacto1080] [--steps 10] [--table-scale 1e-3]

--table-scale multiplies every hash table (and divides the first layers: the same field through small features, what a real checkpoint
looks like -- nerfstudio initialises its tables at 1e-3, tiny-cuda-nn at 1e-4).  r02's normals kernel fell back to exact fp32 there; the
line says which arithmetic ran (sn_effective_precision, kernel 1) and times both.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from signerf_amd import Cameras, scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sheet64", choices=["sheet64", "nerfacto1080"])
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--table-scale", type=float, default=1.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.workload == "sheet64":
    W, H, cfg = 800, 800, scene.benchmark_config(64)
    focal = float(W)
else:
    W, H, cfg = 1920, 1080, scene.proposal_config()
    focal = 1.2 * H
model = cfg.setup()
sd = scene.synthetic_state_dict(cfg, seed=0)
if a.table_scale != 1.0:
    for k in [k for k in sd if k.endswith("encoder.hash_table")]:
        pre = k[: -len("encoder.hash_table")]
        sd[k] = sd[k] * a.table_scale
        sd[pre + "mlp.layers.0.weight"] = sd[pre + "mlp.layers.0.weight"] / a.table_scale
model.load_state_dict(sd, strict=False)
model = model.to(dev).eval()
cam = Cameras(scene.benchmark_cameras(8)[:, :3], focal, focal, W / 2, H / 2, W, H).to(dev)[0]
bundle = cam.generate_rays(camera_indices=0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return sum(s.elapsed_time(e) for s, e in ev) / len(ev)


flat = bundle
from signerf_amd import _lib  # noqa: E402

S = cfg.num_nerf_samples_per_ray
colour = timed(lambda: model._render(flat, H, W))
eff = _lib.load().sn_effective_precision(model._handle, 1, 1)
line = f"{a.workload} {W}x{H}x{S}, tables x{a.table_scale:g}: colour render {colour:.2f} ms"
for prec in ("fp16x2", "fp32"):
    model.config.precision = prec
    t = timed(lambda: model._render_normals(flat, H, W))
    line += f"; normals render precision={prec}: {t:.2f} ms ({W * H * S / t / 1e6:.2f} G ray-samples/s)"
if cfg.num_proposal_iterations > 0:   # r05: the normals launch re-uses the final bins the colour render left in its workspace
    model.config.precision = "fp16x2"

    def both():
        _, st = model._render_ex(flat, H, W, keep_state=True)
        model._render_normals(flat, H, W, st)

    t_both = timed(both)
    t_sep = timed(lambda: (model._render(flat, H, W), model._render_normals(flat, H, W)))
    line += f"; colour + normals of one frame (fp16x2): {t_both:.2f} ms with the colour render's bins re-used, {t_sep:.2f} ms with the proposal sampler run twice"
print(line + f"; a split-precision request resolves to {'fp16x2' if eff == 1 else 'EXACT FP32 (fallback)'} for the normals kernel")
