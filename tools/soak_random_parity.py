#!/usr/bin/env python3
"""One-off soak of the render path against the CPU oracle: N seeded random scenarios drawn from a WIDER space than the suite's fixed
sweeps (tests/test_gpu_random_parity.py) -- sample counts, proposal iterations 0 / 1 / 2, far planes, density levels, frame shapes down
to 1x1, cameras anywhere, render boxes (rays that miss), both initial samplers, contraction on / off with a random scene box, every named
background, both MFMA precisions.  Gate per scenario = the suite's: RMSE <= 1e-3 on rgb / accumulation / (relative) median depth over the
pixels whose reference is finite, identical non-finite pattern, median-depth flips counted (<= 1 in 300).  Failures are printed with
their seed and do not stop the run.

    python tools/soak_random_parity.py --n 150 [--first 0]          (GPU box; the oracle side runs on the host's cores)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import oracle_config, small_config, use_granted_cpus  # noqa: E402

use_granted_cpus()   # the oracle side: as many torch threads as the container is granted (tests/helpers.py)
from oracle import nerfacto as onf  # noqa: E402
from signerf_amd import Cameras, SceneBox, scene  # noqa: E402
from test_gpu_random_parity import _look_at, _random_c2w  # noqa: E402


_TRAINED = {}


def trained_sd():
    """The fitted scene of tools/make_trained_scene.py (r05; fitted once per process, cached on disk)."""
    if "sd" not in _TRAINED:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_trained_scene as mts

        _TRAINED["sd"], _ = mts.trained_state_dict(scene.proposal_config())
    return _TRAINED["sd"]


_SENSITIVE_KEYS = ("rgb", "accumulation", "expected_depth")
_TOTALS = {"pixels": 0, "scenarios": 0, **{who: {k: 0 for k in _SENSITIVE_KEYS} for who in ("hip", "rounded", "exp2")}}


def _outliers(k, got, want, ok):
    """[H,W] bool: pixels whose value is off by more than 1e-2 (depths: relative to max(|want|, 1)) among the gated elements `ok`."""
    e = (got - want).abs()
    if "depth" in k:
        e = e / want.abs().clamp_min(1.0)
    return torch.where(ok, e, torch.zeros_like(e)).amax(-1) > 1e-2


def _oracle_ensemble(sd, ocfg, bundle, n, f, ref, seed, draws=8):
    """The oracle against itself: `draws` renders of the same bundle with every component of the ray origins and directions moved by
    one ulp up or down at random -> per draw, the number of pixels whose rgb / accumulation moved by more than 1e-2."""
    g = torch.Generator().manual_seed(77000 + seed)
    o, d = bundle.origins.cpu(), bundle.directions.cpu()
    out = []
    for _ in range(draws):
        def jiggle(t):
            up = torch.rand(t.shape, generator=g) < 0.5
            return torch.where(up, torch.nextafter(t, torch.full_like(t, float("inf"))), torch.nextafter(t, torch.full_like(t, float("-inf"))))

        r1 = onf.get_outputs_for_camera_ray_bundle(sd, ocfg, jiggle(o), jiggle(d), n, f)
        cnt = {}
        for k in ("rgb", "accumulation"):
            px = (r1[k] - ref[k]).abs().amax(-1)
            cnt[k] = int((torch.where(torch.isfinite(px), px, torch.zeros_like(px)) > 1e-2).sum())
        out.append(cnt)
    return out


def _diagnose_worst_ray(model, bundle, sd, ocfg, out, ref, n, f, W):
    """--inspect on a trained scenario with both proposal nets: WHERE along the chain does the ray of the worst rgb pixel leave the
    oracle?  The instrumented kernels (sn_render_rays_debug) give the positions every stage evaluated and the searchsorted indices; the
    oracle gives its own, plus its densities and weights per level; the HIP field stages are then evaluated AT THE ORACLE'S positions
    (ops.field_forward), which separates a field difference from a placement difference."""
    from signerf_amd import ops

    d = (out["rgb"].cpu() - ref["rgb"]).abs().amax(-1)
    d = torch.where(torch.isfinite(d), d, torch.zeros_like(d))
    r = int(d.flatten().argmax())
    y, x = divmod(r, W)
    o_, d_ = bundle.origins.cpu().reshape(-1, 3), bundle.directions.cpu().reshape(-1, 3)
    nn = None if n is None else n.reshape(-1, 1)
    ff = None if f is None else f.reshape(-1, 1)
    with torch.no_grad():
        dbg = onf.get_outputs(sd, ocfg, o_, d_, nn, ff, return_debug=True)["_debug"]
        dbg1 = onf.get_outputs(sd, ocfg, torch.nextafter(o_, o_ + 1.0), d_, nn, ff, return_debug=True)["_debug"]
    _, dump = ops.render_rays_debug(model, bundle, want=("main_q", "median_index", "prop_q", "pdf_index"))
    torch.set_printoptions(precision=8, linewidth=220, sci_mode=False)
    print(f"-- chain of the worst rgb pixel ({y},{x}) = ray {r}")
    for k in (0, 1):
        hq, oq = dump[f"prop_q_{k}"][r].cpu(), dbg[f"prop_q_{k}"][r]
        print(f"   level {k}: max |q_hip - q_oracle| over its {hq.shape[0]} samples {float((hq - oq).abs().max()):.3e} "
              f"(the oracle against itself with origins + 1 ulp: {float((dbg1[f'prop_q_{k}'][r] - oq).abs().max()):.3e})")
        # the HIP density net of this level AT THE ORACLE'S positions of this ray
        dens = ops.field_forward(model, dbg[f"prop_pos_{k}"][r].reshape(-1, 3).to(model.device), which=k)[0].reshape(-1).cpu()
        od = dbg[f"prop_density_{k}"][r].reshape(-1)
        rel = ((dens - od).abs() / od.abs().clamp_min(1e-30))
        print(f"      density net {k} at the oracle's positions: max relative difference {float(rel.max()):.3e}; oracle densities {od.tolist()}")
        print(f"      oracle weights {dbg[f'prop_weights_{k}'][r].reshape(-1).tolist()}")
        gi, wi = dump[f"pdf_index_{k}"][r].cpu().to(torch.int64), dbg[f"pdf_inds_{k + 1}"][r]
        print(f"      searchsorted indices of resampling step {k}: hip {gi.tolist()}")
        print(f"      {' ' * 43}oracle {wi.tolist()}")
        print(f"      {' ' * 29}oracle, origins + 1 ulp {dbg1[f'pdf_inds_{k + 1}'][r].tolist()}")
    mq, oq = dump["main_q"][r].cpu(), dbg["q"][r]
    print(f"   main field: max |q_hip - q_oracle| {float((mq - oq).abs().max()):.3e} (oracle vs itself + 1 ulp {float((dbg1['q'][r] - oq).abs().max()):.3e}); "
          f"median index hip {int(dump['median_index'][r])} oracle {int(dbg['median_index'].view(-1)[r])}")
    print(f"      oracle main weights {dbg['weights'][r].reshape(-1).tolist()}")
    print(f"      oracle main densities {dbg['density'][r].reshape(-1).tolist()}")
    per = (mq - oq).abs().amax(-1)
    print(f"      per-sample |q_hip - q_oracle| {per.tolist()}")
    # What does ONE quantum (2^-24) of ONE proposal weight do to this pixel?  The oracle on this ray alone, a level's weight i nudged up / down.
    o1, d1 = o_[r:r + 1], d_[r:r + 1]
    n1 = None if nn is None else nn[r:r + 1]
    f1 = None if ff is None else ff[r:r + 1]
    with torch.no_grad():
        base = onf.get_outputs(sd, ocfg, o1, d1, n1, f1)
        for k in (0, 1):
            N = dbg[f"prop_weights_{k}"].shape[1]
            worst = (0.0, None)
            for i in range(N):
                for sgn in (1.0, -1.0):
                    nudge = torch.zeros(1, N, 1)
                    nudge[0, i, 0] = sgn * 2.0 ** -24
                    alt = onf.get_outputs(sd, ocfg, o1, d1, n1, f1, weight_nudge={k: nudge})
                    e = float((alt["rgb"] - base["rgb"]).abs().max())
                    if e > worst[0]:
                        worst = (e, (i, sgn, alt["rgb"][0].tolist(), float(alt["accumulation"][0, 0]), float(alt["depth"][0, 0])))
            print(f"   one quantum (2^-24) on one weight of level {k}: the largest change of this pixel's oracle rgb over the {2 * N} nudges = {worst[0]:.3e}"
                  + ("" if worst[1] is None else f" (sample {worst[1][0]}, sign {worst[1][1]:+.0f}: rgb {worst[1][2]}, accumulation {worst[1][3]:.6g}, depth {worst[1][4]:.6g})"))
    print(f"      for comparison: HIP rgb {out['rgb'].cpu()[y, x].tolist()} oracle rgb {ref['rgb'][y, x].tolist()}")


def scenario(seed, gpu, inspect=(), normals=False, full_tables=False, tcnn=False, lenses=False, fp16=False, trained=False):
    g = torch.Generator().manual_seed(910000 + seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    def ru(lo, hi):
        return float(torch.rand(1, generator=g) * (hi - lo) + lo)

    iters = 0 if normals else ri(0, 2)   # (normals: identical bins on both sides, so the plain gate applies -- tests/test_gpu_normals.py)
    S = [1, 2, 3, 5, 8, 13, 24, 33, 48, 64][ri(0, 9)]
    props = tuple([2, 3, 9, 17, 32, 48, 64, 96, 128][ri(0, 8)] for _ in range(iters))
    sampler = "uniform" if ri(0, 3) == 0 else "piecewise"
    no_contract = ri(0, 3) == 0 and not trained   # (the trained scene was fitted through the contraction)
    full_tables = full_tables or trained
    background = ["last_sample", "last_sample", "white", "black", "random"][ri(0, 4)]
    far = [1000.0, 1000.0, ru(2.0, 60.0), ru(3.0, 8.0)][ri(0, 3)] if sampler == "piecewise" else ru(3.0, 9.0)
    precision = "fp32" if ri(0, 2) == 0 else "fp16x2"
    if fp16:   # r04: the opt-in single-fp16 mode (tiny-cuda-nn checkpoints only) against the oracle's emulation of its roundings (mlp_forward(half=True))
        precision, tcnn = "fp16", True
        if sampler != "piecewise" or no_contract:   # the mode has the default sampler / contraction instantiations only (sn_render_rays says so)
            sampler, no_contract, far = "piecewise", False, (far if far > 9.0 else 1000.0)
    kw = dict(num_nerf_samples_per_ray=S, far_plane=far, proposal_initial_sampler=sampler, disable_scene_contraction=no_contract,
              background_color=background, precision=precision)
    if full_tables:   # nerfacto's own table sizes (T = 2^19 main, 2^17 proposal nets): the shapes the production instantiations are specialised for
        kw.update(log2_hashmap_size=19, proposal_net_args_list=[
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
            {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False}])
    cfg = small_config(num_proposal_iterations=iters, num_proposal_samples_per_ray=props, **kw) if iters else small_config(num_proposal_iterations=0, **kw)
    lo = -1.0 - torch.rand(3, generator=g) * 0.6
    hi = 1.0 + torch.rand(3, generator=g) * 0.6
    sbox = SceneBox(aabb=torch.stack([lo, hi]))
    if tcnn:   # tiny-cuda-nn grid semantics + checkpoint import (oracle/tcnn_layout.py: unpinned); a milder scene than the suite's (gains 1 / 1.5)
        import dataclasses

        from helpers import oracle_params_from_tcnn, synthetic_tcnn_checkpoint

        cfg = dataclasses.replace(cfg, implementation="tcnn", average_init_density=ru(0.3, 4.0))
        ck = synthetic_tcnn_checkpoint(cfg, seed=seed, base_gain=1.0, head_gain=1.5)
        model = cfg.setup(scene_box=sbox)
        model.load_state_dict(ck, strict=False)
        model = model.to(gpu).eval()
        sd = oracle_params_from_tcnn(ck, cfg)
    else:
        # trained: surfaces, empty space, near one-hot proposal weights (cameras land anywhere: inside a sphere, under the ground, outside the sky shell)
        sd = trained_sd() if trained else scene.synthetic_state_dict(cfg, seed=seed, density_bias=ru(0.0, 6.0))
        if trained:
            sd = {k: v for k, v in sd.items() if iters > 0 or not k.startswith("proposal_networks.")}
        model = cfg.setup(scene_box=sbox)
        model.load_state_dict(sd, strict=False)
        model.field.embedding_appearance.embedding.weight.data.copy_(sd["field.embedding_appearance.embedding.weight"])
        model = model.to(gpu).eval()
    ocfg = oracle_config(cfg, scene_aabb=sbox.aabb.tolist())
    if normals:
        import dataclasses

        ocfg = dataclasses.replace(ocfg, predict_normals=True)
    H, W = ri(1, 48), ri(1, 48)
    focal = ru(12.0, 80.0)
    kind = ri(0, 4)
    box = None
    obb = None
    if kind == 0:
        c2w = _random_c2w(g)
    else:
        pos = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * ru(0.3, 1.6)
        c2w = _look_at(pos, (torch.rand(3, generator=g) - 0.5) * 0.5)
        if kind == 3:
            blo = (torch.rand(3, generator=g) - 1.0) * 0.4
            box = SceneBox(aabb=torch.stack([blo, blo + torch.rand(3, generator=g) * 0.6 + 0.05]))
            c2w = _look_at(pos, box.aabb.mean(0))
        if kind == 4:   # the viewer's oriented crop box (Model.get_outputs_for_camera(camera, obb_box)), any rotation
            from signerf_amd import OrientedBox

            obb = OrientedBox(R=_random_c2w(g)[:, :3].contiguous(), T=(torch.rand(3, generator=g) - 0.5) * 0.3, S=torch.rand(3, generator=g) * 0.5 + 0.08)
            c2w = _look_at(pos, obb.T)
    fy_, cx_, cy_ = focal * ru(0.8, 1.25), W / 2 + ru(-2, 2), H / 2 + ru(-2, 2)
    lens, ctype = None, 1
    if lenses:   # r04: the cameras of the original dataset (datasetgenerator.py:274-275): OPENCV distortion, PERSPECTIVE / FISHEYE
        ctype = [1, 1, 2, 3][ri(0, 3)]   # PERSPECTIVE / FISHEYE / EQUIRECTANGULAR (the viewer's three preview types)
        if ri(0, 4) > 0:
            lens = torch.tensor([ru(-0.3, 0.3), ru(-0.1, 0.1), ru(-0.02, 0.02), ru(-0.005, 0.005), ru(-0.01, 0.01), ru(-0.01, 0.01)])
            if ri(0, 5) == 0:
                lens = lens * 6.0    # a lens whose Jacobian degenerates inside the frame: the eps rule of the Newton step
    cams = Cameras(c2w[None], focal, fy_, cx_, cy_, W, H, distortion_params=lens, camera_type=ctype).to(gpu)
    model.render_aabb = box
    if obb is not None:
        bundle = cams[0].generate_rays(camera_indices=0, obb_box=obb)
        out = model.get_outputs_for_camera(cams[0], obb_box=obb)
    else:
        bundle = cams[0].generate_rays(camera_indices=0, aabb_box=box)
        out = model.get_outputs_for_camera_ray_bundle(bundle)
    n = None if bundle.nears is None else bundle.nears.cpu()
    f = None if bundle.fars is None else bundle.fars.cpu()
    ref = onf.get_outputs_for_camera_ray_bundle(sd, ocfg, bundle.origins.cpu(), bundle.directions.cpu(), n, f)
    tag = (f"seed {seed}: {H}x{W}, samples {props}+{S}, far {far:.3g}, {sampler}, box-normalised {no_contract}, {background}, {precision}, "
           f"camera kind {kind}, render box {box is not None}, crop box {obb is not None}")
    problems, msgs = [], []
    sensitive = trained and iters > 0
    if sensitive:
        # the yardstick: the oracle against ITSELF as another correct implementation would differ from it -- the last bit of its
        # exponentials taken another way AND every component of the rays moved by one ulp up or down (the kernels' positions differ from
        # the oracle's by exactly that: max |q - q_oracle| = 6e-8)
        _TOTALS["pixels"] += H * W
        _TOTALS["scenarios"] += 1
        gj = torch.Generator().manual_seed(55000 + seed)

        def jiggle(t):
            up = torch.rand(t.shape, generator=gj) < 0.5
            return torch.where(up, torch.nextafter(t, torch.full_like(t, float("inf"))), torch.nextafter(t, torch.full_like(t, float("-inf"))))

        for mode in ("rounded", "exp2"):
            onf.EXP_MODE = mode
            try:
                r1 = onf.get_outputs_for_camera_ray_bundle(sd, ocfg, jiggle(bundle.origins.cpu()), jiggle(bundle.directions.cpu()), n, f)
            finally:
                onf.EXP_MODE = "torch"
            for k in _SENSITIVE_KEYS:
                okk = torch.isfinite(ref[k]) & torch.isfinite(r1[k])
                if k == "expected_depth":
                    okk = okk & (ref["accumulation"] > 1e-3)
                _TOTALS[mode][k] += int(_outliers(k, r1[k], ref[k], okk).sum())
    if lenses:   # the bundle itself against the oracle's restatement of nerfstudio's ray generation
        rr = onf.generate_rays(c2w[:3], focal, fy_, cx_, cy_, H, W, distortion_params=lens, camera_type=ctype)
        gd, wd = bundle.directions.cpu(), rr["directions"]
        fin = torch.isfinite(wd).all(-1)
        if not torch.equal(torch.isfinite(gd).all(-1), fin):
            problems.append("rays: the non-finite directions differ")
        elif bool(fin.any()):
            e = float((gd[fin] - wd[fin]).abs().max())
            msgs.append(f"rays (type {ctype}, lens {'none' if lens is None else 'yes'}) {e:.1e}")
            # a degenerate lens amplifies the last bit of an early Newton step (chaotic region: |det J| near the 1e-3 rule): counted, not gated by value
            offp = int(((gd - wd).abs().amax(-1)[fin] > (2e-6 if ctype == 2 else 5e-7)).sum())
            if offp > max(1, int(fin.sum()) // 100):
                problems.append(f"rays: {offp} of {int(fin.sum())} directions beyond tolerance (max {e:.2e})")
    if obb is not None:   # the bounds themselves against the oracle's restatement of nerfstudio's intersect_obb
        o, d = bundle.origins.cpu().reshape(-1, 3), bundle.directions.cpu().reshape(-1, 3)
        t0, t1 = onf.intersect_obb(o, d, obb.R, obb.T, obb.S)
        hit = t1 < 1e9
        if not torch.equal(bundle.nears.cpu().reshape(-1) >= 1e9, ~hit):
            # a ray along a face / through an edge may be decided either way by the last ulp of the rotated ray: counted
            nd = int((((bundle.nears.cpu().reshape(-1) >= 1e9) != ~hit)).sum())
            if nd > max(1, hit.numel() // 300):
                problems.append(f"obb: {nd} rays of {hit.numel()} hit / miss differently")
            hit = hit & (bundle.nears.cpu().reshape(-1) < 1e9)
        if bool(hit.any()):
            en = float(((bundle.nears.cpu().reshape(-1)[hit] - t0[hit]).abs() / t0[hit].abs().clamp_min(1e-3)).max())
            ef = float(((bundle.fars.cpu().reshape(-1)[hit] - t1[hit]).abs() / t1[hit].abs().clamp_min(1e-3)).max())
            msgs.append(f"obb bounds {max(en, ef):.1e}")
            if max(en, ef) > 1e-4:
                problems.append(f"obb: bounds off by {max(en, ef):.2e} relative")
    for k in ["rgb", "depth", "accumulation", "expected_depth"] + [f"prop_depth_{i}" for i in range(iters)] + (["normals", "pred_normals"] if normals else []):
        got, want = out[k].cpu(), ref[k]
        ok = torch.isfinite(want)
        if not torch.equal(torch.isfinite(got), ok):
            problems.append(f"{k}: the non-finite pixels differ ({int((torch.isfinite(got) != ok).sum())})")
            continue
        if k == "expected_depth":
            # sum(w t) / (sum(w) + 1e-10) of a nearly empty ray is rounding noise in the REFERENCE too: its alphas 1 - exp(-tau) are quantised to
            # 2^-24, so for accumulation ~1e-6 (a ray grazing the render box) a one-quantum difference of one exp moves the quotient by
            # per cent (seen: 11 % at accumulation 2e-6, both values inside the chunk's clip bounds).  Gated where the ray holds weight.
            ok = ok & (ref["accumulation"] > 1e-3)
        if not bool(ok.any()):
            continue
        d = got[ok].double() - want[ok].double()
        if "depth" in k:
            rel = d.abs() / want[ok].double().abs().clamp_min(1e-6)
            flips = rel > 1e-3
            # (single-fp16 mode: kernel and emulation sum a layer in different orders, a pre-rounding value next to an fp16 boundary lands on the
            #  neighbouring fp16 in one of them -- 2^-11 of one activation -- so whole-bin median flips are ten times as frequent; still counted)
            if k != "expected_depth" and int(flips.sum()) > max(1, int(ok.sum()) // (30 if fp16 else 300)):
                own = None
                if trained:
                    # The trained scene's surfaces put a ray's whole weight on one or two samples: which of two neighbouring samples the
                    # cumulative weight passes 0.5 in (and, one stage earlier, which interval of a near one-hot cdf a resampling u falls
                    # in) hangs on the last bit far more often than in a fog of random weights.  Yardstick, as in tests/test_gpu_trained.py:
                    # the ORACLE against itself with the ray origins moved by one ulp -- flips it produces on its own are conditioning,
                    # not a difference between the two implementations.
                    ref1 = onf.get_outputs_for_camera_ray_bundle(sd, ocfg, torch.nextafter(bundle.origins.cpu(), torch.full((), float("inf"))),
                                                                 bundle.directions.cpu(), n, f)
                    w1 = ref1[k]
                    ok1 = ok & torch.isfinite(w1)
                    own = int(((w1[ok1].double() - want[ok1].double()).abs() / want[ok1].double().abs().clamp_min(1e-6) > 1e-3).sum())
                    msgs.append(f"{k}: {int(flips.sum())} flips of {int(ok.sum())}; the oracle flips {own} against itself under a one-ulp shift of the origins")
                if own is None or int(flips.sum()) > 4 * own:
                    problems.append(f"{k}: {int(flips.sum())} flips of {int(ok.sum())}")
            d = (d / want[ok].double().abs().clamp_min(1.0))[~flips] if k != "expected_depth" else d / want[ok].double().abs().clamp_min(1.0)
        if k in ("normals", "pred_normals"):
            # sum(w n) / (|sum(w n)| + 1e-10) of a nearly empty ray (accumulation ~1e-4: a ray grazing the render box) amplifies the 2^-24
            # quantisation of the reference's own alphas into per-cent changes of the direction: gated where the ray holds weight
            ok = ok & (ref["accumulation"] > 1e-2)
            if not bool(ok.any()):
                continue
            d = got[ok].double() - want[ok].double()
        if k == "normals":
            # the analytic normal is discontinuous where a ReLU of the density MLP switches or a sample sits on a voxel face: a tie decided the
            # other way changes that sample's normal by O(1) (tests/test_gpu_normals.py counts them the same way); pixels, not values
            px = (got - want).abs().amax(-1)[ok.all(-1)]
            ties = px > 1e-3
            if int(ties.sum()) > max(1, px.numel() // 500):
                problems.append(f"{k}: {int(ties.sum())} pixels of {px.numel()} beyond 1e-3")
            d = (got - want)[ok.all(-1)][~ties].double().flatten()
        if sensitive and k in _SENSITIVE_KEYS:
            # r05 (profiles/r05_soak_trained_outlier_chain.txt: seeds 236 and 597 followed stage by stage).  An alpha 1 - exp(-tau) of empty
            # space is a small multiple of 2^-24, so the last bit of exp -- where torch's CPU exp, CUDA's expf and the kernels' v_exp_f32 form all
            # differ on some inputs -- is one QUANTUM of a weight.  In a histogram of few samples that is all padding (sum = N x 0.01) one quantum
            # moves the next level's samples by 1e-6; a trained surface (density x e^22 over 0.01 units) turns that into per cent of a flank
            # sample's weight, the next resampling into 1e-5 .. 1e-3 of position, and one pixel of the frame shows another picture; far samples
            # (t ~ 800) do the same to the expected depth of a nearly empty ray.  Such pixels are counted (<= 1 in 300 per frame, and in total
            # against the ORACLE ITSELF with another exp: oracle/nerfacto.py EXP_MODE, the totals line of the sweep); the rmse gates the others.
            outl = _outliers(k, got, want, ok)
            _TOTALS["hip"][k] += int(outl.sum())
            if 0 < int(outl.sum()) <= max(1, outl.numel() // 300):
                msgs.append(f"{k}: {int(outl.sum())} outlier pixel(s) of {outl.numel()} beyond 1e-2 counted")
                keep = ok & ~outl[..., None]
                d = got[keep].double() - want[keep].double()
                if "depth" in k:
                    d = d / want[keep].double().abs().clamp_min(1.0)
        err = float(torch.sqrt(torch.mean(d ** 2))) if d.numel() else 0.0
        msgs.append(f"{k} {err:.1e}")
        if err > 1e-3:
            problems.append(f"{k}: rmse {err:.2e}")
    if inspect:
        for k in inspect:
            got, want = out[k].cpu(), ref[k]
            d = (got - want).abs().amax(-1)
            d = torch.where(torch.isfinite(d), d, torch.full_like(d, -1.0))
            idx = torch.topk(d.flatten(), min(8, d.numel())).indices
            print(f"-- {k}: worst pixels (y, x): got / want, with near / far / accumulation / depth of the reference")
            for i in idx.tolist():
                y, x = divmod(i, W)
                print(f"   ({y},{x}) got {got[y, x].tolist()} want {want[y, x].tolist()} near {None if n is None else float(n[y, x])} far "
                      f"{None if f is None else float(f[y, x])} acc {float(ref['accumulation'][y, x]):.6g} depth {float(ref['depth'][y, x]):.6g} "
                      f"expected (hip / ref) {float(out['expected_depth'][y, x]):.8g} / {float(ref['expected_depth'][y, x]):.8g}")
        if sensitive:
            print("   the oracle against itself under +-1 ulp of the rays, 48 draws, pixels beyond 1e-2:", _oracle_ensemble(sd, ocfg, bundle, n, f, ref, seed, draws=48))
        if trained and iters == 2 and obb is None and sampler == "piecewise":
            _diagnose_worst_ray(model, bundle, sd, ocfg, out, ref, n, f, W)
        ed = ref["expected_depth"]
        print("   reference expected_depth: min %.6g max %.6g, finite %d of %d" % (float(ed[torch.isfinite(ed)].min()), float(ed[torch.isfinite(ed)].max()),
                                                                                   int(torch.isfinite(ed).sum()), ed.numel()))
    return tag, msgs, problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--full-tables", action="store_true", help="nerfacto's table sizes (2^19 / 2^17) instead of the small ones")
    ap.add_argument("--tcnn", action="store_true", help="tiny-cuda-nn grid semantics and checkpoint import")
    ap.add_argument("--lenses", action="store_true", help="random OPENCV distortion parameters and PERSPECTIVE / FISHEYE / EQUIRECTANGULAR cameras; the bundle is checked too")
    ap.add_argument("--fp16", action="store_true", help="the opt-in single-fp16 mode on tiny-cuda-nn checkpoints against the oracle's emulation of its roundings")
    ap.add_argument("--normals", action="store_true", help="uniform-sampler scenarios only, with the normals kernel's two outputs checked as well")
    ap.add_argument("--trained", action="store_true", help="r05: the TRAINED scene (tools/make_trained_scene.py, full table sizes) instead of random weights")
    ap.add_argument("--inspect", type=int, nargs="*", default=[], help="print the worst pixels of these seeds instead of running the sweep")
    a = ap.parse_args()
    gpu = torch.device("cuda", 0)
    for seed in a.inspect:
        tag, msgs, problems = scenario(seed, gpu, inspect=("expected_depth", "depth", "rgb") + (("normals", "pred_normals") if a.normals else ()), normals=a.normals, full_tables=a.full_tables, tcnn=a.tcnn, lenses=a.lenses, fp16=a.fp16, trained=a.trained)
        print(tag, "|", "; ".join(problems), "|", ", ".join(msgs))
    if a.inspect:
        return
    t0 = time.time()
    bad = 0
    for seed in range(a.first, a.first + a.n):
        try:
            tag, msgs, problems = scenario(seed, gpu, normals=a.normals, full_tables=a.full_tables, tcnn=a.tcnn, lenses=a.lenses, fp16=a.fp16, trained=a.trained)
        except Exception as e:  # noqa: BLE001
            tag, msgs, problems = f"seed {seed}", [], [f"EXCEPTION {type(e).__name__}: {str(e)[:300]}"]
        if problems:
            bad += 1
            print("FAIL", tag, "|", "; ".join(problems), "|", ", ".join(msgs), flush=True)
        else:
            print("ok  ", tag, "|", ", ".join(msgs), flush=True)
    if _TOTALS["scenarios"]:
        t = _TOTALS
        print(f"outlier pixels (beyond 1e-2) over the {t['scenarios']} scenarios behind a proposal sampler, {t['pixels']} pixels: "
              f"HIP vs oracle {t['hip']}; the oracle against itself with its rays moved by +-1 ulp and a correctly rounded exp {t['rounded']}, "
              f"the same with exp2(x log2 e) {t['exp2']}")
        for k in _SENSITIVE_KEYS:
            if t["hip"][k] > 3 * max(t["rounded"][k], t["exp2"][k]) + 3:
                bad += 1
                print(f"FAIL totals: {k}: {t['hip'][k]} outlier pixels against {max(t['rounded'][k], t['exp2'][k])} of the oracle's own")
    print(f"{a.n} scenarios, {bad} with problems, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
