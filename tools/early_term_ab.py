#!/usr/bin/env python3
"""SN_EARLY_TERM=0 vs 1 on the BENCHMARK medium (random-weight scene, density bias +4: no wave ever saturates), interleaved round by round
on one handle, both orders -- the re-run VERDICT r04 "weak 7" asks for: profiles/r04_early_term.txt:3 quoted 1.967 -> 1.861 ms at 640x640
from tests/test_gpu_early_term.py::test_early_termination_pays_on_an_opaque_scene, which times ALL "off" frames first and all "on" frames
second on a freshly created model (clock / allocator warm-up lands on the first setting).  SnRenderOpts.march_stats says how many
wave-steps were skipped (expected: 0).  Prints a small table."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from helpers import make_model  # noqa: E402
from signerf_amd import Cameras, ops, scene  # noqa: E402

dev = torch.device("cuda", 0)
model, _ = make_model(scene.benchmark_config(64), dev)
for size in (640, 800):
    cam = Cameras(scene.benchmark_cameras(8)[:, :3], float(size), float(size), size / 2, size / 2, size, size).to(dev)[0]
    b = cam.generate_rays(camera_indices=0)
    ms = {"0": [], "1": []}
    skipped = {}
    for et in ("0", "1"):
        os.environ["SN_EARLY_TERM"] = et
        ops.reload_env(model)
        _, st = ops.render_with_march_stats(model, b)
        skipped[et] = st["K1"][1] - st["K1"][0]
    for r in range(12):
        for et in (("0", "1") if r % 2 == 0 else ("1", "0")):
            os.environ["SN_EARLY_TERM"] = et
            ops.reload_env(model)
            model.get_outputs_for_camera_ray_bundle(b)
            ev = []
            for _ in range(10):
                a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                model.get_outputs_for_camera_ray_bundle(b)
                c.record()
                ev.append((a, c))
            torch.cuda.synchronize()
            ms[et].append(statistics.median(x.elapsed_time(y) for x, y in ev))
    m0, m1 = statistics.median(ms["0"]), statistics.median(ms["1"])
    print(f"benchmark medium {size}x{size}x64, 12 interleaved rounds of 10 frames: early termination off {m0:.3f} ms, on {m1:.3f} ms "
          f"({(m1 / m0 - 1) * 100:+.2f} %); per-round spread off {min(ms['0']):.3f}-{max(ms['0']):.3f}, on {min(ms['1']):.3f}-{max(ms['1']):.3f}; "
          f"wave-steps skipped: off {skipped['0']}, on {skipped['1']}")
os.environ.pop("SN_EARLY_TERM", None)
