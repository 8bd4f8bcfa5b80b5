#!/usr/bin/env python3
"""The `trained` leg of bench.py: ms per frame of the HIP path on the TRAINED scene of tools/make_trained_scene.py
with the exact early termination on and off, and what fraction of the wave-steps it skips.

    python tools/trained_bench.py [--scene /tmp/scene.pt] [--rounds 8] [--frames 6]

Frames: 800x800 (256 + 96 + 48 samples: the reference sheet's camera through nerfacto's sampler), 1920x1080 (BASELINE configs[3]) and
800x800x64 uniform (BASELINE configs[1]'s sampler on the trained main field).  SN_EARLY_TERM=0 / 1 are interleaved round by round on ONE
handle (sn_debug_reload_env) so that clock and temperature drift hit both alike; HIP events around the render call on one stream.
The scene is fitted here when --scene is missing (torch on the GPU, ~20 s; the oracle's field functions -- test infrastructure: the
timed path is the HIP library alone).  Prints ONE JSON object.
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=None)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--frames", type=int, default=6, help="timed frames per round and setting")
    ap.add_argument("--camera", type=int, default=0)
    args = ap.parse_args()
    from signerf_amd import Cameras, ops, scene

    dev = torch.device("cuda", 0)
    if args.scene and os.path.exists(args.scene):
        blob = torch.load(args.scene, map_location="cpu")
        sd, meta = blob["state_dict"], blob["meta"]
    else:
        import make_trained_scene as mts

        sd, meta = mts.trained_state_dict(scene.proposal_config(), device="cuda")

    def load(cfg):
        m = cfg.setup()
        m.load_state_dict({k: v for k, v in sd.items() if cfg.num_proposal_iterations > 0 or not k.startswith("proposal_networks.")}, strict=False)
        m.field.embedding_appearance.embedding.weight.data.copy_(sd["field.embedding_appearance.embedding.weight"])
        return m.to(dev).eval()

    c2w = scene.benchmark_cameras(8)[:, :3]
    legs = [("800x800, 256 + 96 + 48 samples", scene.proposal_config(), 800, 800, 800.0),
            ("1920x1080, 256 + 96 + 48 samples (BASELINE configs[3]'s shape)", scene.proposal_config(), 1920, 1080, 1.2 * 1080),
            ("800x800, 64 uniform samples, no proposal nets (BASELINE configs[1]'s shape)", scene.benchmark_config(64), 800, 800, 800.0)]
    out = {"scene": {"what": "tools/make_trained_scene.py: torch-path field + proposal nets fitted to an analytic scene (two spheres, ground disc, far sky shell)",
                     **{k: meta[k] for k in ("steps", "points", "device", "seconds") if k in meta}}, "legs": []}
    models = {}
    for name, cfg, W, H, focal in legs:
        key = cfg.num_proposal_iterations
        if key not in models:
            models[key] = load(cfg)
        model = models[key]
        cam = Cameras(c2w, focal, focal, W / 2, H / 2, W, H).to(dev)[args.camera]
        bundle = cam.generate_rays(camera_indices=0)
        ms = {"0": [], "1": []}
        stats = {}
        for et in ("0", "1"):
            os.environ["SN_EARLY_TERM"] = et
            ops.reload_env(model)
            o, st = ops.render_with_march_stats(model, bundle)
            stats[et] = st
            if et == "0":
                ref = {k: o[k].clone() for k in ("rgb", "depth", "accumulation")}
            else:
                same = all(torch.equal(ref[k].nan_to_num(-7.0), o[k].nan_to_num(-7.0)) for k in ref)
                acc = o["accumulation"]
                pic = {"accumulation_mean": float(acc.mean()), "accumulation_above_0.99": float((acc > 0.99).float().mean()),
                       "accumulation_below_0.01": float((acc < 0.01).float().mean()), "rgb_std": float(o["rgb"].std())}
        for r in range(args.rounds):
            for et in (("0", "1") if r % 2 == 0 else ("1", "0")):
                os.environ["SN_EARLY_TERM"] = et
                ops.reload_env(model)
                model.get_outputs_for_camera_ray_bundle(bundle)  # one untimed frame after the switch
                ev = []
                for _ in range(args.frames):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    model.get_outputs_for_camera_ray_bundle(bundle)
                    b.record()
                    ev.append((a, b))
                torch.cuda.synchronize()
                ms[et].append(statistics.median(a.elapsed_time(b) for a, b in ev))
        os.environ.pop("SN_EARLY_TERM", None)
        ops.reload_env(model)
        m0, m1 = statistics.median(ms["0"]), statistics.median(ms["1"])
        S = cfg.num_nerf_samples_per_ray
        n_prop = sum(cfg.num_proposal_samples_per_ray[:cfg.num_proposal_iterations])
        leg = {"frame": name, "ms_per_frame": {"early_term_off": m0, "early_term_on": m1, "speedup": m0 / m1},
               "bit_identical_on_vs_off": bool(same), "picture": pic,
               "field_evaluations_per_s_nominal": W * H * (S + n_prop) / (m1 * 1e-3),
               "wave_steps": {k: {"executed": v[0], "full_march": v[1], "skipped_fraction": 1.0 - v[0] / max(v[1], 1)} for k, v in stats["1"].items()},
               "wave_steps_with_early_term_off": {k: {"executed": v[0], "full_march": v[1]} for k, v in stats["0"].items()},
               "rounds": args.rounds, "frames_per_round": args.frames, "timer": "HIP events around the render call, one stream; median of per-round medians, settings interleaved"}
        out["legs"].append(leg)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
