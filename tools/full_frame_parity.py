#!/usr/bin/env python3
"""Parity of WHOLE frames at BASELINE.json's full sizes against the CPU oracle (one-off; the suite checks crops of these frames,
tests/test_gpu_render.py::test_full_size_properties_*): every pixel of

    sheet64       BASELINE configs[1]: 800 x 800 rays x 64 samples, hash grid L=16 T=2^19, camera 0 of the sheet   (~2.5 min of oracle)
    nerfacto1080  BASELINE configs[3]: 1920 x 1080, proposal nets 256 + 96 samples + 48 main samples               (~15 min of oracle)
    trained800    r05: the TRAINED scene (tools/make_trained_scene.py) through nerfacto's sampler, 800 x 800, 256 + 96 + 48 samples
    trained64     r05: the trained main field behind BASELINE configs[1]'s sampler, 800 x 800 x 64
                  (--crop N: only the centred N x N region of the frame, rendered as its own bundle by both sides)

rendered once by the HIP path (through the C ABI, like everything else) and once by oracle/nerfacto.py in the reference's own chunks of
32 768 rays (signerf_config.py:32; expected_depth is clipped per chunk, A17) on the host cores the box grants.  Reported per output:
RMSE, largest absolute difference, pixels beyond 1e-3, bit-equal pixels; for the depths also the relative error and the number of
median-index flips (a 0.5 crossing decided the other way: the depth jumps by a whole bin).

    python tools/full_frame_parity.py [--only sheet64|nerfacto1080] [--out gpurun_out/full_frame_parity.txt]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import make_model, oracle_config, use_granted_cpus  # noqa: E402

use_granted_cpus()   # the oracle side: as many torch threads as the container is granted (tests/helpers.py)
from oracle import nerfacto as onf  # noqa: E402  (test infrastructure: this tool is a checker, not a product path)
from signerf_amd import Cameras, scene  # noqa: E402


def granted_cpus() -> int:
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    return os.cpu_count() or 1


def compare(name, got, ref, depth_like):
    g, r = got.double().cpu().reshape(-1), ref.double().reshape(-1)
    d = (g - r).abs()
    row = {"output": name, "pixels": int(r.numel()), "rmse": float(torch.sqrt(torch.mean(d * d))), "max_abs": float(d.max()),
           "beyond_1e-3": int((d > 1e-3).sum()), "bit_equal": int((got.cpu().reshape(-1) == ref.reshape(-1)).sum()),
           "non_finite_pattern_equal": bool(torch.equal(torch.isfinite(got.cpu()), torch.isfinite(ref)))}
    if depth_like:
        rel = d / r.abs().clamp_min(1e-30)
        flips = rel > 1e-3
        row["bin_jumps"] = int(flips.sum())
        keep = ~flips
        row["rel_rmse_without_jumps"] = float(torch.sqrt(torch.mean(rel[keep] ** 2))) if bool(keep.any()) else None
        row["rel_max_without_jumps"] = float(rel[keep].max()) if bool(keep.any()) else None
        row["abs_rmse_without_jumps"] = float(torch.sqrt(torch.mean(d[keep] ** 2))) if bool(keep.any()) else None
    return row


def run(workload, dev, camera=0, crop=0):
    if workload in ("sheet64", "trained64"):
        cfg, W, H, focal = scene.benchmark_config(64), 800, 800, 800.0
    elif workload == "trained800":
        cfg, W, H, focal = scene.proposal_config(), 800, 800, 800.0
    else:
        cfg, W, H = scene.proposal_config(), 1920, 1080
        focal = 1.2 * H
    if workload.startswith("trained"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_trained_scene as mts

        sd, _ = mts.trained_state_dict(scene.proposal_config(), device="cuda")
        model = cfg.setup()
        model.load_state_dict({k: v for k, v in sd.items() if cfg.num_proposal_iterations > 0 or not k.startswith("proposal_networks.")}, strict=False)
        model.field.embedding_appearance.embedding.weight.data.copy_(sd["field.embedding_appearance.embedding.weight"])
        model = model.to(dev).eval()
    else:
        model, sd = make_model(cfg, dev)
    cam = Cameras(scene.benchmark_cameras(8)[:, :3], focal, focal, W / 2, H / 2, W, H).to(dev)[camera]
    b = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
    if crop:
        y0, x0 = (H - crop) // 2, (W - crop) // 2
        b = b._map(lambda t: t[y0:y0 + crop, x0:x0 + crop].contiguous())
        W = H = crop
    out = model.get_outputs_for_camera_ray_bundle(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = onf.get_outputs_for_camera_ray_bundle(sd, oracle_config(cfg), b.origins.cpu(), b.directions.cpu())
    secs = time.perf_counter() - t0
    keys = [("rgb", False), ("accumulation", False), ("depth", True), ("expected_depth", True)]
    keys += [(f"prop_depth_{i}", True) for i in range(cfg.num_proposal_iterations)]
    rows = [compare(k, out[k], ref[k], dl) for k, dl in keys]
    return {"workload": workload, "camera": camera, "frame": [W, H], "rays": W * H, "oracle_seconds": secs, "oracle_threads": torch.get_num_threads(),
            "oracle_chunk_rays": cfg.eval_num_rays_per_chunk, "precision": cfg.precision, "outputs": rows,
            "rgb_std": float(ref["rgb"].std()), "accumulation_mean": float(ref["accumulation"].mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=["sheet64", "nerfacto1080", "trained800", "trained64"])
    ap.add_argument("--crop", type=int, default=0, help="compare the centred N x N region only (rendered as its own bundle by both sides)")
    ap.add_argument("--cameras", default="0", help="comma-separated cameras of the 8-camera reference sheet (BASELINE configs[2]: the eight circle_poses views)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_frame_parity.txt"))
    a = ap.parse_args()
    torch.set_num_threads(granted_cpus())
    dev = torch.device("cuda", 0)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        for w, c in [(w, int(c)) for w in ([a.only] if a.only else ["sheet64", "nerfacto1080"]) for c in a.cameras.split(",")]:
            r = run(w, dev, c, a.crop)
            f.write(json.dumps(r) + "\n")
            f.flush()
            print(f"== {w}, camera {c}: {r['frame'][0]} x {r['frame'][1]}, oracle {r['oracle_seconds']:.0f} s on {r['oracle_threads']} threads "
                  f"(rgb std {r['rgb_std']:.3f}, mean accumulation {r['accumulation_mean']:.3f})")
            for row in r["outputs"]:
                extra = (f" | bin jumps {row['bin_jumps']}, rel rmse {row['rel_rmse_without_jumps']:.2e}, rel max {row['rel_max_without_jumps']:.2e}"
                         if "bin_jumps" in row else "")
                print(f"   {row['output']:15s} rmse {row['rmse']:.2e}  max {row['max_abs']:.2e}  beyond 1e-3: {row['beyond_1e-3']:6d}  "
                      f"bit-equal {row['bit_equal']}/{row['pixels']}{extra}")


if __name__ == "__main__":
    main()
