#!/usr/bin/env python3
"""How close is the PER-WAVE exact early termination (csrc/sn_main.h: the march stops once exp(-cumsum(tau)) is exactly 0 for all 64 rays of
an 8x8 tile) to finer exits?  Computed with the ORACLE (test infrastructure; CPU) on the trained scene of tools/make_trained_scene.py: per ray,
the first sample whose front transmittance underflows to 0 in fp32 (v_exp_f32 flushes below 2^-126), then the steps K1 would execute with a
per-ray / per-4x4 / per-half-wave (8x4) / per-wave (8x8) exit, and -- for comparison only, NOT exact, not offered -- with a T < 1e-4 threshold.
VERDICT r04 item 1: "if >= 30 % of tiles straddle a silhouette and never terminate, build the finer criterion".

    python tools/trained_saturation.py [--crop 256]        (~3 min on 8 cores; fits the scene first when it is not cached)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from helpers import oracle_config  # noqa: E402
from oracle import nerfacto as onf  # noqa: E402
from signerf_amd import scene  # noqa: E402
import make_trained_scene as mts  # noqa: E402


def analyse(sd, cfg, W, H, focal, crop, name):
    ocfg = oracle_config(cfg)
    rays = onf.generate_rays(scene.benchmark_cameras(8)[0][:3], focal, focal, W / 2, H / 2, H, W)
    y0, x0 = (H - crop) // 2 // 8 * 8, (W - crop) // 2 // 8 * 8
    o = rays["origins"][y0:y0 + crop, x0:x0 + crop].reshape(-1, 3)
    d = rays["directions"][y0:y0 + crop, x0:x0 + crop].reshape(-1, 3)
    S = cfg.num_nerf_samples_per_ray
    first, first4, acc = [], [], []
    with torch.no_grad():
        for i in range(0, o.shape[0], 4096):
            r = onf.get_outputs(sd, ocfg, o[i:i + 4096], d[i:i + 4096], return_debug=True)
            dbg = r["_debug"]
            eb = dbg["euclid_bins"]
            T = torch.exp(-torch.cumsum((eb[:, 1:] - eb[:, :-1]) * dbg["density"][..., 0], -1))   # transmittance in front of sample i + 1
            for thr, dst in ((2.0 ** -126, first), (1e-4, first4)):
                z = T < thr
                dst.append(torch.where(z.any(-1), z.float().argmax(-1), torch.full((T.shape[0],), S)))
            acc.append(r["accumulation"][:, 0])
    first, first4, acc = (torch.cat(t).view(crop, crop).float() for t in (first, first4, acc))
    tiles = lambda x, th, tw: x.view(crop // th, th, crop // tw, tw).permute(0, 2, 1, 3).reshape(crop // th, crop // tw, -1)  # noqa: E731
    steps = lambda f: torch.clamp(f + 2, max=S)   # noqa: E731  (K1 runs the sample that saturates, then jumps to the last one)
    frac = lambda f, th, tw: float(steps(tiles(f, th, tw).max(-1).values).mean() / S)   # noqa: E731
    t = tiles(first, 8, 8)
    print(f"{name} (centred {crop}x{crop} crop of camera 0, accumulation > 0.99 in {float((acc > 0.99).float().mean()):.1%} of its pixels)")
    print(f"   K1 steps executed / full march, EXACT criterion (T == 0 in fp32): per ray {float(steps(first).mean() / S):.3f} | per 4x4 {frac(first, 4, 4):.3f} | "
          f"per half wave 8x4 {frac(first, 4, 8):.3f} | per wave 8x8 (built) {frac(first, 8, 8):.3f}")
    print(f"   for comparison, NOT exact (T < 1e-4): per ray {float(steps(first4).mean() / S):.3f} | per wave {frac(first4, 8, 8):.3f}")
    print(f"   tiles that never terminate: {float((t.max(-1).values >= S).float().mean()):.1%}; mean spread (max - min) of the first saturated index within a tile: "
          f"{float((t.max(-1).values - t.min(-1).values).mean()):.1f} samples; median per-ray index {float(first.median()):.0f} of {S}")


def analyse_k2(sd, W, H, focal, crop, name):
    """The same question for the two proposal levels of K2 (it stops a level once the transmittance IN FRONT of a sample is 0 for every ray
    of the wave): the oracle's proposal chain, level by level."""
    cfg = scene.proposal_config()
    ocfg = oracle_config(cfg)
    rays = onf.generate_rays(scene.benchmark_cameras(8)[0][:3], focal, focal, W / 2, H / 2, H, W)
    y0, x0 = (H - crop) // 2 // 8 * 8, (W - crop) // 2 // 8 * 8
    o = rays["origins"][y0:y0 + crop, x0:x0 + crop].reshape(-1, 3)
    d = rays["directions"][y0:y0 + crop, x0:x0 + crop].reshape(-1, 3)
    firsts = [[], []]
    with torch.no_grad():
        for i in range(0, o.shape[0], 4096):
            oo, dd = o[i:i + 4096], d[i:i + 4096]
            R = oo.shape[0]
            nears, fars = onf.collider_near_far(R, ocfg)
            weights, sb = None, None
            for lv in range(2):
                n = ocfg.num_proposal_samples_per_ray[lv]
                if lv == 0:
                    sb, eb = onf.initial_sampler(nears, fars, n)
                    sb = sb.expand(R, -1)
                else:
                    sb, _, _ = onf.pdf_sample(sb, weights[..., 0], n, ocfg.histogram_padding)
                    eb = onf.spacing_to_euclidean(sb, nears, fars)
                st, en = eb[:, :-1, None], eb[:, 1:, None]
                dens, _, _, _ = onf.density_field(sd, f"proposal_networks.{lv}.mlp_base", ocfg.proposals[lv], onf.sample_positions(oo, dd, st, en), ocfg.average_init_density)
                weights = onf.get_weights(en - st, dens)
                tau = torch.cumsum(((en - st) * dens)[..., 0], -1)
                z = torch.cat([torch.ones(R, 1), torch.exp(-tau[:, :-1])], 1) < 2.0 ** -126
                firsts[lv].append(torch.where(z.any(-1), z.float().argmax(-1), torch.full((R,), n)))
    print(f"{name} (centred {crop}x{crop} crop of camera 0)")
    for lv in range(2):
        n = ocfg.num_proposal_samples_per_ray[lv]
        f = torch.cat(firsts[lv]).view(crop, crop).float()
        tiles = lambda x, th, tw: x.view(crop // th, th, crop // tw, tw).permute(0, 2, 1, 3).reshape(crop // th, crop // tw, -1)  # noqa: E731
        ex = lambda x: torch.clamp(x + 1, max=n)   # noqa: E731
        print(f"   K2 level {lv} ({n} samples): steps executed / full march: per ray {float(ex(f).mean() / n):.3f} | per half wave 8x4 {float(ex(tiles(f, 4, 8).max(-1).values).mean() / n):.3f} | "
              f"per wave 8x8 (built) {float(ex(tiles(f, 8, 8).max(-1).values).mean() / n):.3f}; tiles that never terminate: {float((tiles(f, 8, 8).max(-1).values >= n).float().mean()):.1%}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=256)
    a = ap.parse_args()
    sd, meta = mts.trained_state_dict(scene.proposal_config(), log=print)
    print("scene:", meta)
    analyse(sd, scene.proposal_config(), 1920, 1080, 1.2 * 1080, a.crop, "1920x1080, 256 + 96 + 48 samples")
    analyse(sd, scene.benchmark_config(64), 800, 800, 800.0, a.crop, "800x800, 64 uniform samples")
    analyse_k2(sd, 1920, 1080, 1.2 * 1080, a.crop, "1920x1080, proposal levels")
    analyse_k2(sd, 800, 800, 800.0, a.crop, "800x800, proposal levels")


if __name__ == "__main__":
    main()
