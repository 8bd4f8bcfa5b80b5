#!/usr/bin/env python3
"""Clock / power / time of K1 variants on one box (the chip sustains ~1.8 GHz of its 2.4 GHz under K1 -- the 26 %
between `roofline.frac` and `frac_at_peak_clock` is power; which design knobs move it?).

Every variant renders the 800x800x64 bench frame back to back for --seconds on one stream while
  * `sn_clock_probe` (one wave on a side stream) counts shader cycles against the constant-rate wall clock  -> sustained GHz,
  * a host thread samples `rocm-smi --showpower --showclocks --json` every 0.2 s                            -> W (socket), sclk as the driver reports it,
  * HIP events time every render call                                                                       -> ms per launch (median).
Variants: the product library; libraries built with other compile-time knobs (signerf_amd/libsignerf_hip_<name>.so, built by
`--build`); environment / precision switches of the product library.  One subprocess per variant (the library and its switches are
fixed at load / create time), interleaved over --rounds.

    python tools/power_ab.py --build          (here: cross-compiles the variant libraries; they travel with the snapshot)
    python tools/power_ab.py [--rounds 2]     (GPU box; prints the table, see profiles/r03_power_ab.txt)
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # docs/history/tools/ -> repository root
sys.path.insert(0, ROOT)

# name -> (extra compile flags | None = product library, environment, precision)
VARIANTS = {
    "base": (None, {}, "fp16x2"),
    "waves2": (("-DSN_MAIN_WAVES_PER_SIMD=2",), {}, "fp16x2"),                 # occupancy: 2 instead of 3 waves per SIMD
    "prio0": (("-DSN_MFMA_PRIO=0",), {}, "fp16x2"),                            # no s_setprio around the MFMA clusters
    "group8": (("-DSN_HASH_GROUP=8",), {}, "fp16x2"),                          # 64 instead of 32 gathers in flight per wave
    "dense0": (None, {"SN_DENSE_LEVELS": "0"}, "fp16x2"),                      # every level hashed: 128 x 8-byte gathers instead of 84 (44 x 16 B + 40 x 8 B)
    "dense8": (None, {"SN_DENSE_LEVELS": "8"}, "fp16x2"),                      # 8 instead of 11 de-hashed levels
    "wide4": (None, {"SN_K1_WIDE": "1"}, "fp16x2"),                            # r04: 8-wave workgroups, FOUR waves per SIMD (tile-sequential MLP, 127 VGPRs)
    "fp32": (None, {}, "fp32"),                                                # exact-fp32 MFMA (320 x 64-cycle MFMAs, no operand splits)
    # DATA variants of the product binary (r04): the same instruction stream on operands that do not toggle -- which part of the package
    # power is data activity, and where?  (early termination off in all four, so that every wave marches all 64 samples)
    "data_base": (None, {"SN_EARLY_TERM": "0"}, "fp16x2"),
    "zero_weights": (None, {"SN_EARLY_TERM": "0", "SN_POWER_SCENE": "zero_weights"}, "fp16x2"),   # MLP matrices 0 (biases kept): A operands of every MFMA are 0
    "zero_table": (None, {"SN_EARLY_TERM": "0", "SN_POWER_SCENE": "zero_table"}, "fp16x2"),       # hash table = one constant: every gather returns the same bits
    "zero_both": (None, {"SN_EARLY_TERM": "0", "SN_POWER_SCENE": "zero_both"}, "fp16x2"),
}


def lib_path(name):
    return os.path.join(ROOT, "signerf_amd", f"libsignerf_hip_{name}.so")


def build():
    from signerf_amd import build as b

    for name, (flags, _, _) in VARIANTS.items():
        if flags:
            print(b.build(extra_flags=flags, out_path=lib_path(name), verbose=False))


def smi_sample():
    """(socket power in W, sclk in MHz) from rocm-smi's JSON, None where it does not report."""
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
    except Exception:  # noqa: BLE001
        return None, None
    power = clock = None
    for k, v in card.items():
        kl = k.lower()
        try:
            if "power" in kl and "(w)" in kl and power is None:
                power = float(v)
            if kl.startswith("sclk clock speed"):
                clock = float(str(v).strip("()").lower().replace("mhz", ""))
        except ValueError:
            pass
    return power, clock


def one(name, seconds):
    import torch

    from signerf_amd import Cameras, _lib, scene

    _, _, precision = VARIANTS[name]
    dev = torch.device("cuda", 0)
    cfg = scene.benchmark_config(64)
    cfg.precision = precision
    model = cfg.setup()
    sd = scene.synthetic_state_dict(cfg, seed=0)
    tweak = os.environ.get("SN_POWER_SCENE", "")
    if tweak in ("zero_weights", "zero_both"):
        for k in sd:
            if k.endswith(".weight") and ".layers." in k:
                sd[k] = torch.zeros_like(sd[k])
    if tweak in ("zero_table", "zero_both"):
        for k in sd:
            if k.endswith("hash_table"):
                sd[k] = torch.full_like(sd[k], 0.5)   # (a constant: nothing toggles on the return path; 0 would defeat the range conditioning)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    cam = Cameras(scene.benchmark_cameras(8)[:, :3], 800.0, 800.0, 400.0, 400.0, 800, 800).to(dev)[0]
    b = cam.generate_rays(0)
    for _ in range(5):
        model.get_outputs_for_camera_ray_bundle(b)
    torch.cuda.synchronize()
    lib = _lib.load()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi_sample())
            time.sleep(0.2)

    th = threading.Thread(target=sampler)
    probe = torch.zeros(3, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    th.start()
    time.sleep(0.3)
    ev = []
    t0 = time.perf_counter()
    t_end = t0 + seconds
    probed = False
    while time.perf_counter() < t_end:
        if not probed and time.perf_counter() - t0 > 0.6 * seconds:
            # the clock of the LAST third of the window: the first second of load still runs at boost clock (r03: 2.37 GHz measured over the
            # first 0.9 s against 1.8-2.0 GHz sustained), so the probe starts once the power controller has settled
            _lib.check(lib.sn_clock_probe(probe.data_ptr(), min(0.9, seconds * 0.3), side.cuda_stream), None, "sn_clock_probe")
            probed = True
        for _ in range(20):
            a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            model.get_outputs_for_camera_ray_bundle(b)
            c.record()
            ev.append((a, c))
        torch.cuda.current_stream().synchronize()   # (NOT the device: the probe wave on the side stream must not stall the render loop)
    torch.cuda.synchronize()
    stop.set()
    th.join()
    cyc, ticks, rate = (int(x) for x in probe.tolist())
    late = sorted(a.elapsed_time(c) for a, c in ev[len(ev) // 2:])      # launches of the second half of the window: settled clock
    ms = sorted(a.elapsed_time(c) for a, c in ev)
    pw = [p for p, _ in samples if p]
    ck = [c for _, c in samples if c]
    print(json.dumps({"name": name, "effective_precision": model.effective_precision, "ms_median": statistics.median(ms), "ms_p05": ms[int(0.05 * len(ms))], "launches": len(ms), "ms_late": statistics.median(late),
                      "probe_ghz": cyc / (ticks / rate) / 1e9 if ticks > 0 and rate > 0 else None,
                      "power_w": statistics.median(pw) if pw else None, "power_w_max": max(pw) if pw else None,
                      "smi_sclk_mhz": statistics.median(ck) if ck else None, "smi_samples": len(samples)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--one")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--only", default=None, help="comma-separated variant names (default: all)")
    a = ap.parse_args()
    if a.build:
        return build()
    if a.one:
        return one(a.one, a.seconds)
    rows = {}
    for rnd in range(a.rounds):
        for name, (flags, env, _) in VARIANTS.items():
            if a.only and name not in a.only.split(","):
                continue
            e = dict(os.environ, **env)
            if flags:
                if not os.path.exists(lib_path(name)):
                    print(f"{name}: variant library missing (run --build first)")
                    continue
                e["SIGNERF_HIP_LIB"] = lib_path(name)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, "--seconds", str(a.seconds)], env=e, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(f"{name}: failed: {r.stderr[-400:]}")
                continue
            rows.setdefault(name, []).append(json.loads(line[-1]))
    try:
        cap = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout
        print("\n".join(ln for ln in cap.splitlines() if "Max" in ln or "max" in ln))
    except Exception:  # noqa: BLE001
        pass
    print("variant   ms/launch (2nd half of window)   probe GHz   socket W   W x ms (mJ per frame)   smi sclk MHz   what")
    what = {"base": "product library, fp16x2", "waves2": "2 waves per SIMD", "prio0": "no s_setprio around MFMA clusters", "group8": "64 gathers in flight",
            "dense0": "no de-hashed copies: 128 hashed gathers", "dense8": "8 de-hashed levels", "fp32": "exact-fp32 MFMA",
            "wide4": "8-wave workgroups, 4 waves per SIMD, tile-sequential MLP (SN_K1_WIDE=1)",
            "data_base": "product library, early termination off", "zero_weights": "same binary, MLP matrices = 0 (MFMA A operands do not toggle)",
            "zero_table": "same binary, hash table = one constant (gathers return the same bits)", "zero_both": "same binary, matrices 0 and table constant"}
    for name, rs in rows.items():
        med = lambda k: statistics.median([r[k] for r in rs if r.get(k) is not None]) if any(r.get(k) is not None for r in rs) else float("nan")  # noqa: E731
        ms, ghz, w = med("ms_late"), med("probe_ghz"), med("power_w")
        print(f"{name:8s}  {ms:7.3f}                        {ghz:6.3f}     {w:7.1f}    {w * ms:8.1f}                {med('smi_sclk_mhz'):7.0f}        {what.get(name, '')} [{rs[0].get('effective_precision')}]")


if __name__ == "__main__":
    main()
