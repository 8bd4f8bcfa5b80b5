#!/usr/bin/env python3
"""bench.py -- ray-samples/s and ms/frame of the reference-sheet render path on MI355X.

    python bench.py --gpus 1 --steps 300 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = every rank renders ONE 800x800 camera of the 3x3 reference sheet (BASELINE.json configs[1]: synthetic nerfacto
field, hash grid L=16 T=2^19, 64 samples/ray, no proposal nets; rank r renders camera r of circle_poses(8)) through
Cameras.generate_rays -> Model.get_outputs_for_camera_ray_bundle, followed (N>1) by the RCCL all-gather of the finished
[H,W,4] tiles (started asynchronously: it overlaps the next step's render; all K gathers complete inside the timed region).
Weak scaling: per-GPU work is fixed.  Inputs (weights, camera) are resident in HBM before the timed region.
Prints ONE JSON line (see DESIGN.md "Measurement"):
  value            whole-job ray-samples/s over the K timed steps (wall clock between barriers + device syncs, max over ranks)
                   Consecutive steps go to --frames-in-flight (default 2) alternating HIP streams: frames are independent, so the head of
                   one fills the wave slots the tail of the previous leaves idle (800x800 = 3.26 rounds of waves; signerf_amd.sheet.FrameStreams)
  kernel_ms        HIP-event time of a render call alone on the chip (one-stream leg after the timed region): mean / median / min / max / p05 / p95
  roofline         the hardware issue roof that binds the dominant kernel (K1 sn_render_main_kernel), as a fraction < 1: instruction
                   counts per wave-step come from the disassembly of the loaded library (tools/kernel_counts.py), the clock from
                   sn_clock_probe running beside the renders; `roofs` lists every roof considered
  roofline_hbm     the SURVEY §8(d) line: algorithmic bytes (1024 B per main-field sample) over the 8 TB/s HBM peak.  It exceeds 1
                   because the 64 MiB table is served by L1/L2/Infinity Cache; `traffic` = fabric bytes per launch from the rocprofv3
                   PMC passes committed under profiles/ (labelled with the commit they were taken at)
  cpu_baseline     the CPU oracle (a port of nerfstudio's torch fallback) on bounded samples of configs 2, 1 and 4, rank 0, N = 1
"""
import argparse
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
PEAK_CLOCK_GHZ = 2.4            # MI355X_MICROARCH.md: peak engine clock (the chip sustains 1.75-1.85 GHz under these kernels: power)
BYTES_PER_MAIN_SAMPLE = 1024.0  # 16 levels x 8 corners x 2 features x 4 B
N_CUS, N_SIMDS = 256, 1024      # MI355X_MICROARCH.md chip-level parameters
VALU_ISSUE_CYCLES = 4.0         # (r01-r05's flat price of a VALU / MFMA issue slot; r06: only the default of an opcode the table below does not list)
# r06 -- what ONE wave64 instruction costs the vector issue port of a SIMD, per opcode, measured (tools/probes/valu_issue_probe.hip,
# profiles/r06_valu_issue_probe.txt: 1-8 waves per SIMD of independent instructions of one kind, no memory, no MFMA).  Three classes:
#   2 cycles  v_fma_f32 / v_fmac / v_mul / v_sub / v_add_f32, v_add_u32, v_xor, v_and, v_or, v_bitop3, v_mov    (2.2-2.6 measured with >= 3 waves: the
#             guide's "v_fma_f32 (wave64) 2 cyc (SIMD-32)"; ONE wave alone issues one per 4.4-5.4 cycles, which is where r02's 3.86 came from)
#   4 cycles  v_cvt_pkrtz_f16_f32, v_fma_mix_f32, v_pk_max_f16, v_pk_fma_f32, v_mad_u32_u24, v_mul_u32_u24, v_max_i32, v_cvt_i32_f32,
#             v_fract_f32, v_lshlrev_b32, v_max3_f32, v_add_f64, v_cvt_f64_f32   (4.1-4.4 measured)
#   8 cycles  v_rcp_f32, v_exp_f32, v_permlane32_swap  (8.1-8.3);   v_cndmask_b32 with a VCC mask: 19.5 (!)
# The roof prices every class at its NOMINAL cost (2 / 4 / 8 / 20) -- the hardware's best, not the 3-wave figure -- and an MFMA's issue at 4.
ISSUE_CYCLES = {
    2.0: ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_sub_f32", "v_subrev_f32", "v_add_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32",
          "v_or_b32", "v_bitop3_b32", "v_mov_b32", "v_mov_b64", "v_accvgpr_read_b32", "v_accvgpr_write_b32"),
    8.0: ("v_rcp_f32", "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_permlane32_swap_b32", "v_permlane16_swap_b32"),
    20.0: ("v_cndmask_b32",),
}
ISSUE_CYCLES_OF = {op: c for c, ops in ISSUE_CYCLES.items() for op in ops}


def issue_port_cycles(opcodes):
    """(vector-issue-port cycles per wave-step, {class: instructions}) of an opcode histogram (tools/kernel_counts.py `opcodes`): listed
    opcodes at their class price, every other VALU opcode and every MFMA at 4 cycles."""
    total, by = 0.0, {}
    for op, n in opcodes.items():
        c = 4.0 if op.startswith(("v_mfma", "v_smfma")) else ISSUE_CYCLES_OF.get(op, VALU_ISSUE_CYCLES)
        total += c * n
        k = "mfma issue (4)" if op.startswith(("v_mfma", "v_smfma")) else "%g-cycle" % c
        by[k] = by.get(k, 0) + n
    return total, by
GATHER_MIN_CYCLES = 16.0        # a 64-lane gather costs the CU's L1/TA >= 16 cycles (4 lanes per clock), DESIGN.md "What the L1 charges"
MFMA_F16_CYCLES = 32.0          # v_mfma_f32_32x32x16_f16 pipe time per SIMD
MFMA_F32_CYCLES = 64.0          # v_mfma_f32_32x32x2_f32
MFMA_PORT_CYCLES = 14.0         # vector-issue-port time an f16 32x32x16 MFMA takes in the probes (12.4-16.9; its operand traffic), for the empirical model


# The driver's record (BENCH_rNN.json `parsed.roofline`) keeps the LEADING scalar keys of `roofline` only -- r05's record ended after
# 21 of them, strings cut at 128 characters, nested dicts dropped (VERDICT r05 "weak 4").  So the figures a reader needs beside `value`
# come first, flat, in this order; everything else (notes, per-roof tables, nested legs) follows.  tests/test_bench_host.py holds the order.
ROOFLINE_LEADING_KEYS = (
    "bound", "achieved", "peak", "unit", "frac", "traffic",
    "kernel_ms", "one_launch_ms", "one_launch_ray_samples_per_s",
    "exact_fp32_ms", "exact_fp32_ray_samples_per_s",
    "hbm_algorithmic_ratio", "traffic_over_algorithmic", "traffic_frac_of_hbm_peak", "l2_hit_rate",
    "mfma_frac", "sustained_clock_ghz",
    "configs3_ms_per_frame", "configs4_ms_per_view", "trained_800_ms", "T21_ms",
    "simd_issue_frac", "l1_gather_issue_frac", "frac_at_sustained_clock",
)


def flat_roofline(rf, line):
    """`rf` (the roofline dict built during the run) re-ordered for the driver: ROOFLINE_LEADING_KEYS first -- every one present, a scalar
    or None -- then the remaining keys of `rf` in their order.  Values are looked up in `rf` itself, then derived from the other
    sections of the line (`alt_precision`, `roofline_hbm`, `roofline_mfma`, `others`)."""
    rf = dict(rf)
    alt = line.get("alt_precision") or {}
    me_fp32 = str(line.get("dtype", "")).startswith("f32")
    hbm = line.get("roofline_hbm") or {}
    mfma = line.get("roofline_mfma") or {}
    d = {
        "one_launch_ms": rf.get("kernel_ms"),
        "exact_fp32_ms": rf.get("kernel_ms") if me_fp32 else (alt.get("kernel_ms") if alt.get("precision") == "fp32" else None),
        "exact_fp32_ray_samples_per_s": (rf.get("one_launch_ray_samples_per_s") if me_fp32 else
                                         (alt.get("ray_samples_per_s_per_gpu") if alt.get("precision") == "fp32" else None)),
        # SURVEY 8(d)'s line, named for what it is: algorithmic bytes per second over the 8 TB/s HBM peak.  A RATIO, not a fraction of a
        # binding roof (the table is served by L2 / Infinity Cache: `traffic_over_algorithmic`)
        "hbm_algorithmic_ratio": hbm.get("frac"),
        "traffic_over_algorithmic": hbm.get("traffic_over_algorithmic"),
        "traffic_frac_of_hbm_peak": hbm.get("traffic_frac_of_hbm_peak"),
        "mfma_frac": mfma.get("frac"),
    }
    for leg in line.get("others") or []:
        c = leg.get("config", "")
        if "error" in leg:
            continue
        if c.startswith("BASELINE.json configs[3]"):
            d["configs3_ms_per_frame"] = leg.get("ms_per_frame")
        elif c.startswith("BASELINE.json configs[4]"):
            d["configs4_ms_per_view"] = leg.get("ms_per_view")
        elif c.startswith("trained scene"):
            for x in leg.get("legs") or []:
                if str(x.get("frame", "")).startswith("800x800") and "256" in str(x.get("frame", "")):
                    d["trained_800_ms"] = (x.get("ms_per_frame") or {}).get("early_term_on")
        elif c.startswith("BASELINE configs[1] frame with T = 2^"):
            d["T21_ms"] = leg.get("kernel_ms_per_launch")
    out = {}
    for k in ROOFLINE_LEADING_KEYS:
        v = rf.get(k)
        if v is None:
            v = d.get(k)
        out[k] = v if isinstance(v, (int, float, str)) or v is None else None
    for k, v in rf.items():
        if k not in out:
            out[k] = v
    return out


def measured_traffic(precision):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json, written by
    tools/pmc_summary.py --json: TCC_EA0_RDREQ_{32,64,128}B + WRITE_SIZE).  bench.py cannot run rocprofv3 on itself, so the entry
    carries the commit it was profiled at and the line says so."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f).get(precision, {})
            return e.get("bytes_per_launch"), e.get("commit", "r01 (ae42c63)")
    except (OSError, ValueError):
        return None, None


def _json_line_of(cmd, timeout_s, env=None, cwd=None):
    """Runs a child process and returns the last JSON object it printed on a line of its own (None + reason on any failure)."""
    import subprocess

    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=cwd)
    except Exception as e:  # noqa: BLE001
        return None, repr(e)
    for ln in reversed(r.stdout.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                return json.loads(ln), None
            except ValueError:
                continue
    # (tools/config5_bench.py pretty-prints one object)
    try:
        i = r.stdout.index("{")
        return json.loads(r.stdout[i:]), None
    except ValueError:
        return None, "rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])


def other_configs(precision):
    """BASELINE.json configs[3] and configs[4] in the driver's line: untimed legs AFTER the headline's timed region, each in
    a child process of this interpreter so that the headline's handle, streams and allocator state are not disturbed.
      configs[3]  `bench.py --workload nerfacto1080 --steps 12 --warmup 3`: 1920x1080, proposal nets 256 + 96 + 48 main samples
      configs[4]  `tools/config5_bench.py --size 800 --only-nopng`: DatasetGenerator.generate_dataset, 8 reference + 50 views, PNG writes off
      + (r04) `tools/tcnn_modes_bench.py`: the headline frame on a tiny-cuda-nn grid, fp32-grade default vs the opt-in single-fp16 mode"""
    out = []
    me = os.path.abspath(__file__)
    base = [sys.executable, me, "--workload", "nerfacto1080", "--steps", "12", "--warmup", "3", "--precision", precision, "--no-cpu-baseline",
            "--no-alt-precision", "--no-others", "--no-traffic"]
    d, err = _json_line_of(base, 200)
    if d is None:
        out.append({"config": "BASELINE.json configs[3] (nerfacto1080)", "error": err})
    else:
        rf = d.get("roofline", {})
        out.append({"config": d["config"]["workload"], "command": "bench.py --workload nerfacto1080 --steps 12 --warmup 3 (child process, after the headline)",
                    "ms_per_frame": d["ms_per_step"], "frames": d["steps"], "frames_in_flight": d.get("frames_in_flight"),
                    "kernel_ms_per_launch": d["kernel_ms"]["median"], "ray_samples_per_s": d["value"],
                    "field_evaluations_per_s": d.get("field_evaluations_per_sec"), "rays_per_s": d.get("rays_per_sec"),
                    "roofline": {k: rf.get(k) for k in ("bound", "frac", "frac_at_sustained_clock", "sustained_clock_ghz", "peak_clock_ghz", "unit")}})
    cmd = [sys.executable, os.path.join(ROOT, "tools", "config5_bench.py"), "--size", "800", "--reps", "1", "--only-nopng"]
    d, err = _json_line_of(cmd, 200)
    if d is None:
        out.append({"config": "BASELINE.json configs[4] (generator loop)", "error": err})
    else:
        e = d.get("png_writes_off", {})
        out.append({"config": "BASELINE.json configs[4]: DatasetGenerator.generate_dataset, " + d.get("workload", ""), "size": d.get("size"),
                    "command": "tools/config5_bench.py --size 800 --reps 1 --only-nopng (child process; one warm-up repetition + one timed)",
                    "png_writes": "off", "total_ms": e.get("total_ms"), "ms_per_view": e.get("ms_per_view"),
                    "render_stage_ms": e.get("render_stage_ms"), "render_ms_per_view": e.get("render_ms_per_view"),
                    "field_evaluations_per_s": e.get("field_evaluations_per_s"), "views": d.get("views"), "ranks": d.get("ranks")})
    # the same frame on a tiny-cuda-nn grid, fp32-grade default and the opt-in single-fp16 mode (what the library itself computes with): timing only
    d, err = _json_line_of([sys.executable, os.path.join(ROOT, "tools", "tcnn_modes_bench.py"), "--rounds", "12"], 150)
    out.append({"config": "tiny-cuda-nn grid (SURVEY 8(f) row 2), 800x800x64: fp32-grade default vs the opt-in single-fp16 mode", "error": err} if d is None else
               dict({"config": "tiny-cuda-nn grid (SURVEY 8(f) row 2), 800x800x64: fp32-grade default vs the opt-in single-fp16 mode (NOT fp32-grade, never the headline)",
                     "command": "tools/tcnn_modes_bench.py --rounds 12 (child process)"}, **d))
    out.append(trained_leg())
    out.append(big_table_leg(precision))
    return out


def trained_leg():
    """r05: the TRAINED scene (tools/make_trained_scene.py: the torch-path field + proposal nets fitted to an analytic scene with surfaces and
    empty space -- the regime every real SIGNeRF render is in, README.md:146,170).  `tools/trained_bench.py` in a child process: ms per frame
    with the exact early termination on / off (interleaved), the fraction of wave-steps it skips per kernel (SnRenderOpts.march_stats)."""
    d, err = _json_line_of([sys.executable, os.path.join(ROOT, "tools", "trained_bench.py"), "--rounds", "6", "--frames", "5"], 420)
    if d is None:
        return {"config": "trained scene (tools/make_trained_scene.py)", "error": err}
    return dict({"config": "trained scene (tools/make_trained_scene.py): HIP render with the exact early termination on / off",
                 "command": "tools/trained_bench.py --rounds 6 --frames 5 (child process; fits the scene with torch on the GPU first, untimed)"}, **d)


def big_table_leg(precision, log2_t=21):
    """r05: the headline frame with a 2^21-row table per level -- the library's stated limit (include/signerf_hip.h):
    256 MiB of table + its de-hashed copies no longer fit the 256 MiB Infinity Cache, so SURVEY 8(d)'s NAMED roof (HBM) can bind and the
    128-byte request for an 8-byte row becomes visible.  A child `bench.py --log2-hashmap-size 21` for the timing + three in-run rocprofv3
    PMC passes (TCC_EA0_RDREQ_*, WRITE_SIZE, TCC_HIT / TCC_MISS) of the same command."""
    me = os.path.abspath(__file__)
    extra = ["--log2-hashmap-size", str(log2_t)]
    d, err = _json_line_of([sys.executable, me, "--steps", "12", "--warmup", "3", "--precision", precision, "--no-cpu-baseline", "--no-alt-precision",
                            "--no-others", "--no-traffic", *extra], 200)
    name = f"BASELINE configs[1] frame with T = 2^{log2_t} rows per level ({(16 << log2_t) * 8 >> 20} MiB table)"
    if d is None:
        return {"config": name, "error": err}
    k_ms = d["kernel_ms"]["median"]
    alg = d["roofline_hbm"]["algorithmic_bytes_per_launch"]
    tr, terr = inrun_traffic(precision, extra_args=extra, with_l2=True)
    rf = d.get("roofline", {})
    leg = {"config": name, "command": "bench.py --log2-hashmap-size %d --steps 12 --warmup 3 (child) + rocprofv3 --pmc passes of the same" % log2_t,
           "ms_per_frame": d["ms_per_step"], "kernel_ms_per_launch": k_ms, "ray_samples_per_s": d["value"],
           "simd_issue_frac": rf.get("frac"), "simd_issue_frac_at_sustained_clock": rf.get("frac_at_sustained_clock"), "sustained_clock_ghz": rf.get("sustained_clock_ghz"),
           "algorithmic_bytes_per_launch": alg, "traffic": tr, "traffic_error": terr}
    if tr:
        gbps = tr["bytes_per_launch"] / (k_ms * 1e-3) / 1e9
        leg.update({"traffic_bytes_per_launch": tr["bytes_per_launch"], "traffic_gbps": gbps, "traffic_frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
                    "traffic_over_algorithmic": tr["bytes_per_launch"] / alg, "l2_hit_rate": tr.get("l2_hit_rate")})
        hbm, issue = gbps / HBM_PEAK_GBPS, rf.get("frac") or 0.0
        leg["binding_roof"] = "hbm" if hbm > issue else "simd-issue"
        leg["note"] = ("fabric bytes (Infinity-Cache hits included: an upper bound of DRAM bytes) over the 8 TB/s HBM peak vs the vector issue port's "
                       "busy fraction at the 2.4 GHz peak clock; the larger one binds")
    return leg


def inrun_traffic(precision, kernel_pattern="sn_render_main_kernel", extra_args=(), with_l2=False):
    """HBM-side bytes per launch of K1 from rocprofv3 PMC passes collected IN THIS RUN (MI355X_MICROARCH.md "HBM": the L2's memory-side
    request counters, separate --pmc passes, never combined with a system trace): a child `bench.py --steps 3` per pass under
    `rocprofv3 --kernel-trace --pmc ...`.  Reads = sum over size classes of TCC_EA0_RDREQ_{32,64,128}B x size (the size-classed counters need
    no gfx950 halving correction: that applies to FETCH_SIZE, which tallies 128-B requests at 64 B); writes = WRITE_SIZE KiB (the guide calls
    it uncalibrated; it is < 1 % of the total here).  Infinity-Cache hits are counted, so this is fabric traffic, an upper bound of DRAM bytes."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="sn_traffic_")
    env = dict(os.environ, TMPDIR="/tmp")
    me = os.path.abspath(__file__)
    child = [sys.executable, me, "--steps", "3", "--warmup", "1", "--frames-in-flight", "1", "--precision", precision, "--no-cpu-baseline",
             "--no-alt-precision", "--no-others", "--no-traffic", *extra_args]
    acc = {}
    passes = [("rd", ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]), ("wr", ["WRITE_SIZE"])]
    if with_l2:   # L2 hit rate of the same dispatches (its own pass, as the guide prescribes)
        passes.append(("l2", ["TCC_HIT_sum", "TCC_MISS_sum"]))
    try:
        for name, counters in passes:
            import subprocess

            d = os.path.join(tmp, name)
            r = subprocess.run([exe, "--kernel-trace", "--output-format", "csv", "--pmc", *counters, "-d", d, "--", *child],
                               capture_output=True, text=True, timeout=150, env=env, cwd="/tmp")
            rows = 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_pattern in row.get("Kernel_Name", ""):
                        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                        rows += 1
            if rows == 0:
                return None, "pass %s: no counter rows (rc %d) %s" % (name, r.returncode, (r.stderr or "")[-200:])
        a = {k: sum(v) / len(v) for k, v in acc.items()}
        rd = 32 * a.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * a.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * a.get("TCC_EA0_RDREQ_128B_sum", 0)
        wr = a.get("WRITE_SIZE", 0) * 1024
        out = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launches_averaged": len(next(iter(acc.values()))),
               "read_requests": {k: a[k] for k in sorted(a) if k.startswith("TCC_EA0_RDREQ")}}
        if with_l2 and a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 0) > 0:
            out["l2_hit_rate"] = a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
        return out, None
    except Exception as e:  # noqa: BLE001
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def device_identity(dev):
    """What tells two GPUs of a node apart, as short strings: name, PCI bus id, UUID, compute units (torch's device properties)."""
    pr = torch.cuda.get_device_properties(dev)
    bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    return {"name": pr.name, "pci": bus, "uuid": str(getattr(pr, "uuid", "")), "cus": pr.multi_processor_count,
            "arch": getattr(pr, "gcnArchName", "")}


def preflight(world, rank, dev, backend, shared_gpu):
    """N > 1, BEFORE the first render (the first real 8-GPU run should be boring): every rank reports who it is --
    rank, device index, PCI bus id, UUID -- over the process group itself (the first collective of the run: if the backend cannot even do
    this, the job fails here with a clear text and not inside the timed region), and the facts are CHECKED: the world size the backend
    sees, one distinct device per rank (RCCL), one library version everywhere.  Returns flat fields for `config` (rank 0) -- the driver's
    record keeps scalars."""
    rccl = None
    if backend == "nccl":
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            rccl = "unknown (%s)" % type(e).__name__
    me = dict(device_identity(dev), rank=rank, device_index=dev.index, pid=os.getpid(), rccl=rccl, torch=torch.__version__)
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    # HARD failures: what cannot be true of a working process group (a wrong answer here would make every later number meaningless)
    assert [e["rank"] for e in everyone] == list(range(world)), "all_gather_object returned ranks out of order"
    assert dist.get_world_size() == world, "the process group does not see WORLD_SIZE ranks"
    # one tiny device-side collective as well: the sum of the ranks (catches a fabric that moves objects over the host but not tensors)
    t = torch.tensor([float(rank)], device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    assert float(t.item()) == world * (world - 1) / 2, "all_reduce of the rank numbers returned %r" % float(t.item())
    # SOFT findings (reported in the line, never fatal: a property this torch build does not expose must not cost the measurement)
    distinct = len({(e["pci"], e["uuid"], e["device_index"]) for e in everyone})
    notes = []
    if backend == "nccl" and not shared_gpu and distinct != world:
        notes.append("ranks share a device identity: %s" % [(e["rank"], e["pci"], e["device_index"]) for e in everyone])
    if len({e["rccl"] for e in everyone}) != 1 or len({e["torch"] for e in everyone}) != 1:
        notes.append("ranks run different RCCL / torch builds")
    return {"preflight": "ok" if not notes else ("warning: " + "; ".join(notes))[:120], "rccl_version": me["rccl"], "distinct_devices": distinct,
            "device_name": me["name"], "device_cus": me["cus"],
            "devices": "; ".join("r%d %s" % (e["rank"], e["pci"]) for e in everyone)[:120]}


def rank_spread(value, world, dev, backend):
    """min / max / mean / arg-max over the ranks of one per-rank number (the one-launch render time: a slow die shows here)."""
    t = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t)
    v = t.tolist()
    return {"min": min(v), "max": max(v), "mean": sum(v) / len(v), "slowest_rank": max(range(world), key=lambda i: v[i]), "per_rank": v}


def physical_cores():
    """(physical cores, logical CPUs) of the host."""
    pairs, phys, core = set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip() and phys is not None and core is not None:
                    pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    n = len(pairs) or max(1, logical // 2)
    return min(n, logical), logical


def run_with_watchdog(fn, timeout_s, on_timeout):
    """fn() with a watchdog: if it has not returned after timeout_s seconds, on_timeout() runs on a timer thread (it is expected to end the
    process -- a hung collective cannot be cancelled).  A call that returns in time never sees on_timeout."""
    done = threading.Event()

    def fire():
        if not done.is_set():
            on_timeout()

    dog = threading.Timer(timeout_s, fire)
    dog.daemon = True
    dog.start()
    try:
        return fn()
    finally:
        done.set()
        dog.cancel()


def cpu_quota():
    """CPUs the container may actually use: the cgroup CPU quota (v2 cpu.max, v1 cfs_quota / cfs_period), None when unlimited.  A GPU box of
    this pool shows 256 logical CPUs and grants 16 (measured r04: host threads stop scaling there, docs/history/tools/png_scaling_probe.py)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, per = float(f.read()), float(g.read())
            return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(main_cfg, main_sd, width, height, samples, runs=5, runs_config4=3):
    """The CPU oracle (kind "port": nerfstudio itself cannot be installed here) timed on the bounded samples BASELINE.md §2 names
    (SURVEY §8(d)): a centred 200x200 crop of the headline frame (config 2), config 1 in full, a centred 240x135 crop of config 4;
    `time.perf_counter` around the render, one warm-up, MEDIAN of the timed renders -- BASELINE.md's 5 for configs 2 and 1 (r06; ~8 s and
    ~0.1 s each on the GPU box's 16 granted cores), 3 for the config-4 crop (~45 s each: five of them would put the default run past five
    minutes; `--cpu-runs-config4 5` gives the plan's count).  The CPU leg of a default run: ~3 min.  Threads = the cores the container is
    granted, set explicitly."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_config, small_config
    from oracle import nerfacto as onf
    from signerf_amd import scene

    cores, logical = physical_cores()
    quota = cpu_quota()
    physical = cores
    if quota is not None:   # more threads than granted CPUs only get throttled
        cores = max(1, min(cores, int(quota + 0.5)))
    old_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except (OSError, StopIteration):
        pass

    def timed(cfg, sd, W, H, focal, crop_w, crop_h, n_runs, cam=0, chunk=None):
        c2w = scene.benchmark_cameras(8)[cam]
        rays = onf.generate_rays(c2w[:3], focal, focal, W / 2, H / 2, H, W)
        y0, x0 = (H - crop_h) // 2, (W - crop_w) // 2
        o = rays["origins"][y0:y0 + crop_h, x0:x0 + crop_w].contiguous()
        d = rays["directions"][y0:y0 + crop_h, x0:x0 + crop_w].contiguous()
        ocfg = oracle_config(cfg)
        wh = min(16, crop_h)
        onf.get_outputs_for_camera_ray_bundle(sd, ocfg, o[:wh].contiguous(), d[:wh].contiguous(), chunk=chunk)  # warm-up: threads, allocator, code paths
        times = []
        for _ in range(max(1, n_runs)):
            t = time.perf_counter()
            onf.get_outputs_for_camera_ray_bundle(sd, ocfg, o, d, chunk=chunk)
            times.append(time.perf_counter() - t)
        return statistics.median(times), times

    crop = 200
    dt2, all2 = timed(main_cfg, main_sd, width, height, float(width), crop, crop, runs)
    out = {"value": crop * crop * samples / dt2, "unit": "ray-samples/s", "cores": cores, "threads": cores, "logical_cpus": logical, "physical_cores_of_the_host": physical,
           "cgroup_cpu_quota": quota,
           "kind": "port", "host_cpu": cpu, "timer": "time.perf_counter around the render, 1 warm-up + median of the timed renders",
           "sample": f"config 2: centred {crop}x{crop} crop of the {width}x{height}x{samples} frame, median of {len(all2)} renders = {dt2:.1f} s, "
                     "torch CPU fp32 oracle (BASELINE.md section 2: 1 warm-up + median of 5); SAMPLE_OTHERS",
           "runs": len(all2), "seconds_each": all2,
           "rays_per_s": crop * crop / dt2,
           "ms_per_frame_extrapolated": dt2 * 1e3 * (width * height) / (crop * crop), "others": []}
    c1 = small_config(num_proposal_iterations=0, num_nerf_samples_per_ray=32)
    dt1, all1 = timed(c1, scene.synthetic_state_dict(c1, seed=0), 64, 64, 64.0, 64, 64, max(runs, 5))
    out["others"].append({"config": "config 1: 64x64 image, 32 samples/ray, in full", "runs": len(all1), "seconds": dt1, "seconds_each": all1,
                          "rays_per_s": 64 * 64 / dt1, "ray_samples_per_s": 64 * 64 * 32 / dt1, "ms_per_frame": dt1 * 1e3})
    c4 = scene.proposal_config()
    sd4 = scene.synthetic_state_dict(c4, seed=0)
    cw, ch = 240, 135
    dt4, all4 = timed(c4, sd4, 1920, 1080, 1.2 * 1080, cw, ch, runs_config4, chunk=8192)  # chunked: bounds the oracle's memory (352 proposal samples per ray)
    n4 = cw * ch
    out["others"].append({"config": f"config 4: centred {cw}x{ch} crop of the 1920x1080 frame, proposal nets 256 + 96 + 48 main samples",
                          "runs": len(all4), "seconds": dt4, "seconds_each": all4, "rays_per_s": n4 / dt4,
                          "ray_samples_per_s": n4 * 48 / dt4, "field_evaluations_per_s": n4 * 400 / dt4,
                          "ms_per_frame_extrapolated": dt4 * 1e3 * (1920 * 1080) / n4})
    out["sample"] = out["sample"].replace("SAMPLE_OTHERS", f"also timed: config 1 in full (64x64x32), median of {len(all1)} renders = {dt1:.2f} s; "
                                          f"config 4: centred {cw}x{ch} crop of the 1920x1080 frame (256 + 96 + 48 samples), "
                                          f"{'median of ' + str(len(all4)) + ' renders' if len(all4) > 1 else 'ONE render'} = {dt4:.1f} s (`others`)")
    torch.set_num_threads(old_threads)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default 300: >= 0.9 s of render at ~3 ms per frame)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="sheet64", choices=["sheet64", "nerfacto1080"],
                    help="sheet64 = BASELINE.json configs[1] (the metric's configuration, default); "
                         "nerfacto1080 = configs[3]: 1920x1080, 2 proposal nets (256 + 96 samples) + 48 main samples")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--log2-hashmap-size", type=int, default=None,
                    help="rows per level of the main hash table (default: nerfacto's 19; 21 = the library's limit, a 256 MiB table that leaves the Infinity Cache)")
    ap.add_argument("--precision", default="fp16x2", choices=["fp32", "fp16x2"],
                    help="MFMA arithmetic of the tiny MLPs: fp16x2 = fp32 operands split into fp16 hi+lo, fp32 accumulate "
                         "(error equals exact fp32 MFMA inside the validated operand range, tests/test_gpu_precision.py); fp32 = exact fp32 MFMA")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1: nccl = RCCL over xGMI (the real thing); gloo = host-staged gathers, for dry "
                         "runs of the N > 1 code on a box with fewer GPUs than ranks (ranks then share GPUs)")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="consecutive steps are issued on this many alternating HIP streams (signerf_amd.sheet.FrameStreams: frames of "
                         "different cameras are independent, the head of one fills the wave slots the tail of the previous leaves idle); "
                         "1 = one stream, every launch waits for the previous one to drain")
    ap.add_argument("--no-alt-precision", action="store_true", help="skip the extra (untimed-region) run of the other precision")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the untimed legs of BASELINE.json configs[3] and configs[4] (`others` in the line)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed CPU-oracle renders of configs 2 and 1 (median reported; BASELINE.md §2: 5)")
    ap.add_argument("--cpu-runs-config4", type=int, default=3, help="timed CPU-oracle renders of the 240x135 config-4 crop (~45 s each; BASELINE.md §2 plans 5)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the driver's command): every rank renders ONE camera per step.  strong: a step is the whole "
                         "8-camera reference sheet of BASELINE.json configs[2] (datasetgenerator.py:517-519), camera i -> rank i mod N, "
                         "one tile gather per sheet; total work is fixed, `ms_per_step` is ms per sheet")
    ap.add_argument("--gather", default="all", choices=["all", "root"],
                    help="N > 1: all-gather the tiles (every rank holds the sheet; north_star's collective) or gather them to rank 0 "
                         "only (the rank that composes the sheet and talks to the diffuser; 1/N of the bytes)")
    ap.add_argument("--diagnostics-timeout", type=float, default=120.0,
                    help="N > 1: seconds the per-strategy exposed-gather legs (after the timed region) may take before the line is printed without them")
    ap.add_argument("--gather-strategy", default="all_gather", choices=["all_gather", "p2p", "all_to_all"],
                    help="N > 1: how the tiles travel (signerf_amd.sheet.gather_tiles_async): RCCL's all-gather / gather, direct point-to-point "
                         "pushes (one per xGMI link), or the same pushes as one all_to_all_single; the line reports the exposed time of ALL three")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    n_dev = torch.cuda.device_count()
    shared_gpu = world > n_dev
    if shared_gpu and args.backend == "nccl":
        sys.exit(f"{world} ranks on {n_dev} GPU(s): RCCL needs one GPU per rank (use --backend gloo for a dry run)")
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from signerf_amd import Cameras, _lib, build, scene, sheet

    # The library ships prebuilt with the snapshot (__graft_entry__.build()).  Only a missing file is rebuilt, by local rank 0
    # alone -- N ranks recompiling into one path at the same time would race.
    if not os.path.exists(build.LIB_PATH):
        if local_rank == 0:
            build.build(verbose=False)
        if world > 1:
            dist.barrier()
    if args.workload == "sheet64":
        args.width, args.height, args.samples = args.width or 800, args.height or 800, args.samples or 64
        cfg = scene.benchmark_config(args.samples)
        focal = float(args.width)
        bytes_per_ray = args.samples * BYTES_PER_MAIN_SAMPLE
    else:
        args.width, args.height = args.width or 1920, args.height or 1080
        cfg = scene.proposal_config()
        args.samples = cfg.num_nerf_samples_per_ray
        focal = 1.2 * args.height
        bytes_per_ray = sum(cfg.num_proposal_samples_per_ray) * 320.0 + args.samples * BYTES_PER_MAIN_SAMPLE  # 161.8 kB
    cfg.precision = args.precision
    if args.log2_hashmap_size:
        cfg.log2_hashmap_size = args.log2_hashmap_size
    sd = scene.synthetic_state_dict(cfg, seed=0)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    W, H, S = args.width, args.height, args.samples
    cams = Cameras(scene.benchmark_cameras(8)[:, :3], focal, focal, W / 2, H / 2, W, H).to(dev)
    cam = cams[rank % 8]

    render_ms = []

    def kernel_ms_of(precision: str, n: int = 10) -> float:
        """Mean HIP-event time of the render call in another arithmetic mode (outside the timed region)."""
        old = model.config.precision
        model.config.precision = precision
        ev = []
        for i in range(n + 1):
            bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            model.get_outputs_for_camera_ray_bundle(bundle)
            b.record()
            if i > 0:
                ev.append((a, b))
        torch.cuda.synchronize()
        model.config.precision = old
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)

    def sustained_clock_ghz(frames: int = 12) -> float:
        """Shader clock the chip sustains under this render: sn_clock_probe's eight waves (one per XCD) on a side stream count shader cycles against
        the constant-rate wall clock while `frames` renders run on the main stream."""
        lib = _lib.load()
        ms = statistics.median(a.elapsed_time(b) for a, b in render_ms) if render_ms else 5.0
        out = torch.zeros(3, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        _lib.check(lib.sn_clock_probe(out.data_ptr(), min(0.9, max(0.005, frames * ms * 1e-3 * 0.8)), side.cuda_stream), None, "sn_clock_probe")
        for _ in range(frames):
            model.get_outputs_for_camera_ray_bundle(cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb))
        torch.cuda.synchronize()
        cyc, ticks, rate = (int(x) for x in out.tolist())
        return cyc / (ticks / rate) / 1e9 if ticks > 0 and rate > 0 else float("nan")

    pending = [None]  # the previous step's tile all-gather, still in flight

    frames = sheet.FrameStreams(dev, max(1, args.frames_in_flight))
    issued = [0]

    def step(timed: bool):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        handle = None
        with frames.frame(issued[0]):  # this step's stream (one of --frames-in-flight, round robin)
            bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
            e0.record()
            out = model.get_outputs_for_camera_ray_bundle(bundle)
            e1.record()
            tile = frames.keep(torch.cat([out["rgb"], out["depth"]], dim=-1)[None])
            if world > 1:
                # depth-1 pipeline: this frame's all-gather (RCCL's own stream, ordered behind this step's stream) overlaps the next
                # frames' renders; every gather is waited for inside the timed region (drain() below)
                handle = sheet.gather_tiles_async(tile, world, dst=dst, strategy=args.gather_strategy)
        issued[0] += 1
        if timed:
            render_ms.append((e0, e1))
        if world == 1:
            return tile
        done = pending[0].wait() if pending[0] is not None else None  # on the caller's stream, not on a render stream
        pending[0] = handle
        return done

    n_sheet = 8                                   # BASELINE.json configs[2]: the 3x3 sheet's eight reference cameras
    strong = args.scaling == "strong"
    dst = 0 if (args.gather == "root" and world > 1) else None
    mine = sheet.shard_indices(n_sheet, world, rank)

    def sheet_step(timed: bool):
        """--scaling strong: one whole sheet.  This rank renders its cameras (i mod N == rank) on the alternating streams, the caller's
        stream joins them, ONE gather moves the tiles; the gather of sheet k overlaps the renders of sheet k + 1 (depth-1 pipeline)."""
        tiles = []
        for i in mine:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with frames.frame(issued[0]):
                bundle = cams[i].generate_rays(camera_indices=0, aabb_box=model.render_aabb)
                e0.record()
                out = model.get_outputs_for_camera_ray_bundle(bundle)
                e1.record()
                tiles.append(frames.keep(torch.cat([out["rgb"], out["depth"]], dim=-1)))
            issued[0] += 1
            if timed:
                render_ms.append((e0, e1))
        frames.join()                             # the caller's stream (where the gather is issued) waits for this sheet's renders
        local = torch.stack(tiles) if tiles else torch.zeros((0, H, W, 4), dtype=torch.float32, device=dev)
        if world == 1:
            return local
        handle = sheet.gather_tiles_async(local, n_sheet, dst=dst, strategy=args.gather_strategy)
        done = pending[0].wait() if pending[0] is not None else None
        pending[0] = handle
        return done

    def drain():
        if pending[0] is not None:
            tiles = pending[0].wait()
            pending[0] = None
            return tiles

    def exposed_gather_ms(n: int = 5, strategy: str = "all_gather"):
        """The tile exchange on its own (N > 1): renders complete, then the gather is issued and waited for at once -- what every step
        would pay if the exchange were not overlapped with the next renders.  HIP events on the caller's stream."""
        ev = []
        for _ in range(n + 1):
            if strong:
                tl = []
                for i in mine:
                    o_ = model.get_outputs_for_camera_ray_bundle(cams[i].generate_rays(camera_indices=0, aabb_box=model.render_aabb))
                    tl.append(torch.cat([o_["rgb"], o_["depth"]], dim=-1))
                local = torch.stack(tl) if tl else torch.zeros((0, H, W, 4), dtype=torch.float32, device=dev)
                n_items = n_sheet
            else:
                o_ = model.get_outputs_for_camera_ray_bundle(cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb))
                local, n_items = torch.cat([o_["rgb"], o_["depth"]], dim=-1)[None], world
            torch.cuda.synchronize()
            dist.barrier()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sheet.gather_tiles_async(local, n_items, dst=dst, strategy=strategy).wait()
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return statistics.median(x.elapsed_time(y) for x, y in ev[1:])

    model._ensure_engine()  # library load, weight upload and de-hashed copies are set-up, not a step (matters only for --warmup 0)
    torch.cuda.synchronize()
    pre = preflight(world, rank, dev, args.backend, shared_gpu) if world > 1 else {"preflight": "n/a (one rank)", "device_name": device_identity(dev)["name"]}
    step_fn = sheet_step if strong else step
    if args.warmup < args.frames_in_flight:
        # every render stream allocates its output / workspace blocks on first use (hipMalloc): one priming frame per stream is set-up
        # like the upload above (matters only for --warmup 0 / 1)
        for _ in range(max(1, args.frames_in_flight)):
            step_fn(False)
    for _ in range(args.warmup):
        step_fn(False)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tiles = step_fn(True)
    tiles = drain() if world > 1 else tiles
    frames.join()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        if dst is None or rank == dst:
            assert tiles is not None and tuple(tiles.shape) == (n_sheet if strong else world, H, W, 4), "the gathered sheet must hold every tile"
        else:
            assert tiles is None
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    latency = sorted(a.elapsed_time(b) for a, b in render_ms)  # with several frames in flight: the span of a call that shared the chip
    # Per-launch time of the render call with nothing else on the chip: an untimed-region leg on ONE stream (what rocprofv3 reports for
    # `bench.py --frames-in-flight 1`, profiles/); the roofline fractions below are per launch and use this.
    if args.frames_in_flight > 1:
        serial = []
        for i in range(min(args.steps, 100) + 6):  # (the first calls on this stream allocate their buffers afresh)
            bundle = cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            model.get_outputs_for_camera_ray_bundle(bundle)
            b.record()
            if i >= 6:
                serial.append((a, b))
        torch.cuda.synchronize()
        per_step = sorted(a.elapsed_time(b) for a, b in serial)
        render_ms[:] = serial  # (the clock probe sizes itself from these)
    else:
        per_step = latency
    kernel_ms = sum(per_step) / max(len(per_step), 1)
    gather_ms = gather_err = None
    gather_by_strategy = {}
    spread = tiles_check = None
    if world > 1:
        # (every rank; after the timed region) the ranks' own one-launch times, and the exchange checked for CONTENT: one more gather of
        # "camera r from rank r", each tile compared bit for bit with this rank's own render of that camera -- the kernels are
        # deterministic, so a tile that differs was damaged on the way (or a peer's GPU computes differently)
        my_kernel_ms = statistics.median(per_step)   # (taken now: rank 0 re-uses the name `per_step` further down)

        def post_checks():
            """Three collectives, executed by every rank WHATEVER happens to its local work in between (a rank that raised between two of
            them would leave the others waiting in the next one): local failures travel as sentinel values instead."""
            local_err = None
            sp = rank_spread(my_kernel_ms, world, dev, args.backend)
            try:
                o_ = model.get_outputs_for_camera_ray_bundle(cam.generate_rays(camera_indices=0, aabb_box=model.render_aabb))
                mine_tile = torch.cat([o_["rgb"], o_["depth"]], dim=-1)[None]
            except Exception as e:  # noqa: BLE001
                local_err, mine_tile = repr(e)[:120], torch.zeros((1, H, W, 4), dtype=torch.float32, device=dev)
            got = sheet.gather_tiles_async(mine_tile, world, dst=None, strategy="all_gather").wait()
            bad = 0
            try:
                for r in range(world):
                    o_ = model.get_outputs_for_camera_ray_bundle(cams[r % 8].generate_rays(camera_indices=0, aabb_box=model.render_aabb))
                    want = torch.cat([o_["rgb"], o_["depth"]], dim=-1)
                    bad += 0 if torch.equal(got[r].to(want.device), want) else 1
            except Exception as e:  # noqa: BLE001
                local_err, bad = local_err or repr(e)[:120], world
            t = torch.tensor([float(bad)], device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t)
            out = {"tiles_checked_per_rank": world, "mismatching_tiles_all_ranks": int(t.item())}
            if local_err:
                out["local_error_rank_%d" % rank] = local_err
            return sp, out

        def run_post_checks(line=None):
            """Under a watchdog, like the gather diagnostics: a check that hangs must not cost the line.  Rank 0 calls this AFTER its line
            is built (the other ranks wait for it inside the first collective): if the limit passes, it prints the line as it stands and
            every rank leaves with status 0 -- the measurement itself is complete."""
            def bail():
                if line is not None:
                    line["config"]["gathered_tiles_bit_identical"] = "post-run checks did not finish within their time limit"
                    print(json.dumps(line), flush=True)
                os._exit(0)

            try:
                return run_with_watchdog(post_checks, args.diagnostics_timeout if rank == 0 else args.diagnostics_timeout + 240.0, bail)
            except Exception as e:  # noqa: BLE001  (diagnostics must not cost the line)
                return None, {"error": repr(e)[:120]}

    def gather_diagnostics(line=None):
        """The exposed cost of the tile gather, per strategy: diagnostic legs AFTER the timed region.  They must never cost the line -- an
        exception is recorded per strategy, and a collective that HANGS (p2p / all_to_all have only ever run on gloo and on one-rank RCCL) is cut
        by a watchdog: rank 0 prints the line it has (gather_ms: error) and every rank leaves with status 0, the measurement being complete."""
        by, err = {}, [None]

        def bail():
            if line is not None:
                line["gather_ms"] = {"exposed": None, "error": "the diagnostic gather legs did not finish within their time limit", "exposed_by_strategy": by}
                print(json.dumps(line), flush=True)
            os._exit(0)

        def legs():
            for strat in sheet.GATHER_STRATEGIES:
                try:
                    by[strat] = exposed_gather_ms(strategy=strat)
                except Exception as e:  # noqa: BLE001
                    by[strat] = None
                    err[0] = repr(e)

        run_with_watchdog(legs, args.diagnostics_timeout if rank == 0 else args.diagnostics_timeout + 240.0, bail)   # (rank 0 builds its line first)
        return by, err[0]

    if world > 1 and rank != 0:
        run_post_checks()
        gather_diagnostics()
    if rank == 0:
        n_steps = len(per_step)
        pct = lambda q: per_step[min(n_steps - 1, int(q * n_steps))]  # noqa: E731
        k_med = statistics.median(per_step)
        cams_per_step = n_sheet if strong else world          # cameras rendered by the whole job in one step
        samples_per_step = cams_per_step * W * H * S
        value = samples_per_step * args.steps / elapsed
        line = {
            "metric": "ray-samples/sec (%dx%d reference-sheet camera render)" % (W, H),
            "value": value, "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32 (exact fp32 MFMA)" if args.precision == "fp32" else
                     "fp16x2-split multiply (22-bit operand significands: every fp32 MLP operand carried as an fp16 hi+lo pair on the matrix "
                     "cores, products hi.hi + hi.lo + lo.hi, lo.lo dropped), f32 accumulate; hash-grid blend, sampling and compositing in f32/f64.  "
                     "Layers range-conditioned into fp16's [2^-3, 65504] by exact power-of-two scales at sn_finalize_weights, exact-fp32 MFMA "
                     "fallback otherwise; error equals exact fp32's (tests/test_gpu_precision.py); the exact-fp32 figure is `alt_precision`",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[1]: {W}x{H} rays x {S} samples/ray, nerfacto hash grid L=16 T=2^{cfg.log2_hashmap_size} F=2, "
                                    "no proposal nets, random-weight synthetic scene, one camera per GPU + tile all-gather")
                       if args.workload == "sheet64" else
                       (f"BASELINE.json configs[3]: {W}x{H} rays, proposal nets 256 + 96 samples (L=5, T=2^17) + {S} main samples "
                        "(L=16, T=2^19), random-weight synthetic scene, one camera per GPU + tile all-gather"),
                       "rays_per_gpu": W * H, "samples_per_ray": S,
                       "step": ("one 8-camera reference sheet (configs[2]; datasetgenerator.py:517-519), camera i -> rank i mod N, one tile gather"
                                if strong else "one camera per rank + tile gather"),
                       "cameras_per_step": cams_per_step,
                       "parallelism": f"camera-sharded x{world}" + (f", tile {'gather to rank 0' if dst is not None else 'all-gather'} overlapped "
                                                                    "with the next renders" if world > 1 else ""),
                       "backend": (args.backend if world > 1 else None), "ranks_share_a_gpu": shared_gpu,
                       # what the process group itself reports (an N > 1 line must show the backend saw N ranks)
                       "dist_world_size": dist.get_world_size() if world > 1 else 1,
                       "dist_backend": dist.get_backend() if world > 1 else None,
                       "cuda_device_count": n_dev, "gather": (args.gather if world > 1 else None),
                       "gather_strategy": (args.gather_strategy if world > 1 else None),
                       # r06 preflight / post-run checks (flat: the driver's record keeps scalars)
                       **pre,
                       # (filled in by the post-run checks below, once the line's own measurements are in it)
                       "rank_kernel_ms_min": None, "rank_kernel_ms_max": None, "rank_kernel_ms_mean": None, "slowest_rank": None,
                       "gathered_tiles_bit_identical": None},
            "rank_kernel_ms": None, "gathered_tiles_check": None,
            "ms_per_frame": elapsed / args.steps * 1e3 / (len(mine) if strong else 1) if (not strong or mine) else None,
            "ms_per_sheet": (elapsed / args.steps * 1e3) if strong else None,
            "gather_ms": None,   # (world > 1: filled in below, after the line's own measurements)
            "timed_region_s": elapsed,
            "frames_in_flight": max(1, args.frames_in_flight),
            # for comparison with one-launch-at-a-time figures (r01's lines, rocprofv3): samples of one frame over the per-launch time
            "one_stream_ray_samples_per_s_per_gpu": W * H * S / (k_med * 1e-3),
            # SURVEY 8(d) metric (1) as written: samples of one frame over the HIP-event time of ONE get_outputs_for_camera_ray_bundle call
            # with nothing else in flight (median); `value` above is the job's pipelined rate (frames_in_flight)
            "value_one_launch_at_a_time": (W * H * S / (k_med * 1e-3)) if world == 1 else None,
            "kernel_ms": {"mean": kernel_ms, "median": k_med, "min": per_step[0], "max": per_step[-1], "p05": pct(0.05), "p95": pct(0.95), "n": n_steps,
                          "what": "HIP-event time of a render call on its launch stream with nothing else in flight (K1 + two memsets + the clip "
                                  "kernel; + K2 with proposal nets)" + ("; measured in a one-stream leg after the timed region -- the timed steps "
                                  "overlap the tail of one frame with the head of the next, see frame_latency_ms" if args.frames_in_flight > 1 else "")},
            "frame_latency_ms": {"median": statistics.median(latency), "p05": latency[int(0.05 * len(latency))], "p95": latency[int(0.95 * len(latency))],
                                 "what": "HIP-event span of each TIMED render call on its own stream (with %d frames in flight a call shares the "
                                         "chip with its neighbours: the span grows, the frame period ms_per_step shrinks)" % max(1, args.frames_in_flight)},
            "rays_per_sec": cams_per_step * W * H * args.steps / elapsed,
            # SURVEY §8(d) metric (3): every field evaluation of a ray (proposal nets + main field)
            "field_evaluations_per_sec": cams_per_step * W * H * (S + (sum(cfg.num_proposal_samples_per_ray[:cfg.num_proposal_iterations])
                                                               if args.workload != "sheet64" else 0)) * args.steps / elapsed,
        }
        achieved_gbps = (W * H * bytes_per_ray) / (k_med * 1e-3) / 1e9
        traffic, traffic_commit = measured_traffic(args.precision) if args.workload == "sheet64" else (None, None)
        # HBM-side bytes per launch: measured IN THIS RUN by two rocprofv3 --pmc child passes (inrun_traffic; N = 1, the headline workload);
        # when that is skipped or fails `traffic` is null and `traffic_source` says so.  The figure of the committed PMC passes is quoted
        # beside it, labelled with the commit it was profiled at.
        line["traffic_committed_profile"] = {"bytes_per_launch": traffic, "profiled_at": traffic_commit,
                                             "source": "profiles/traffic.json (rocprofv3 --pmc TCC_EA0_RDREQ_* / TCC_EA0_WRREQ_*, tools/pmc_summary.py)"}
        inrun, inrun_err = (None, "skipped (--no-traffic)" if args.no_traffic else "N > 1 or not the headline workload")
        if world == 1 and args.workload == "sheet64" and not args.no_traffic:
            inrun, inrun_err = inrun_traffic(args.precision, extra_args=(["--log2-hashmap-size", str(args.log2_hashmap_size)] if args.log2_hashmap_size else []),
                                             with_l2=True)
        traffic_now = inrun["bytes_per_launch"] if inrun else None
        traffic_source = ("in-run rocprofv3 --pmc passes (child `bench.py --steps 3 --frames-in-flight 1`): TCC_EA0_RDREQ_{32,64,128}B_sum x size "
                          "+ WRITE_SIZE KiB, averaged over the K1 dispatches") if inrun else "not measured in this run (%s); see traffic_committed_profile" % inrun_err
        line["traffic_in_run"] = inrun
        line["roofline_hbm"] = {
            "bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved_gbps / HBM_PEAK_GBPS,
            "traffic": traffic_now, "traffic_source": traffic_source,
            "traffic_gbps": (traffic_now / (k_med * 1e-3) / 1e9) if traffic_now else None,
            "traffic_frac_of_hbm_peak": (traffic_now / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic_now else None,
            "traffic_over_algorithmic": (traffic_now / (W * H * bytes_per_ray)) if traffic_now else None,
            "kernel": ("sn_render_main_kernel<0,%d>" % (0 if args.precision == "fp32" else 1)) if args.workload == "sheet64"
            else "sn_proposal_kernel + sn_render_main_kernel<1,*> (whole render call)",
            "kernel_ms": k_med, "algorithmic_bytes_per_launch": W * H * bytes_per_ray,
            "note": "SURVEY 8(d)'s definition (1024 algorithmic bytes per main-field sample).  frac > 1 is NOT a fraction of a binding "
                    "roof: the 64 MiB table never leaves L1/L2/Infinity Cache (`traffic_committed_profile`: fabric bytes per launch from "
                    "the committed rocprofv3 PMC passes, ~0.13x the algorithmic bytes).  The roofs that bind are in `roofline`."}
        if args.workload == "sheet64":
            # The issue roofs of K1.  Per wave-step (64 samples) the kernel issues a fixed instruction mix -- counted from the
            # disassembly of the library that is loaded -- and every resource below serves one such instruction per so many cycles.
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import kernel_counts

                cnt = kernel_counts.loop_counts("sn_render_main_kernelILi0ELi%dELi0ELi11E" % (0 if args.precision == "fp32" else 1))
            except Exception as e:  # noqa: BLE001  (no llvm-objdump: the roofs are then unavailable, the throughput line is not)
                cnt = {"error": repr(e)}
            clock = sustained_clock_ghz()
            if "valu" in cnt and clock == clock:
                wave_steps = W * H * S / 64.0
                t_kernel = k_med * 1e-3
                mfma_cyc = MFMA_F32_CYCLES if args.precision == "fp32" else MFMA_F16_CYCLES
                n_gather = cnt.get("gather", cnt["vmem_load"])
                port_cycles, port_classes = issue_port_cycles(cnt.get("opcodes") or {"v_unknown": cnt["valu"], "v_mfma": cnt["mfma"]})
                roofs = {
                    "simd-issue": {"per_wave_step": cnt["valu"] + cnt["mfma"], "cycles_each": port_cycles / (cnt["valu"] + cnt["mfma"]), "units": N_SIMDS,
                                   "port_cycles_per_wave_step": port_cycles, "instructions_by_class": port_classes,
                                   "what": "VALU + MFMA instructions through the one vector issue port of a SIMD, each opcode at its measured class price "
                                           "(2 / 4 / 8 / 20 cycles per wave64 instruction; profiles/r06_valu_issue_probe.txt); cycles_each = the mix's mean"},
                    "l1-gather-issue": {"per_wave_step": n_gather, "cycles_each": GATHER_MIN_CYCLES, "units": N_CUS,
                                        "what": "64-lane buffer_load gathers through the CU's texture-address / L1 path"},
                    "matrix-pipe": {"per_wave_step": cnt["mfma"], "cycles_each": mfma_cyc, "units": N_SIMDS,
                                    "what": "MFMA instructions through the SIMD's matrix pipe"},
                }
                for r in roofs.values():
                    r["achieved"] = r["per_wave_step"] * wave_steps / t_kernel / 1e9                  # G instructions / s, whole chip
                    # `frac` / `peak`: against the PART's peak (2.4 GHz), as a roofline fraction is quoted; the same roof priced at the
                    # clock the chip sustains under this kernel (power-limited) is `frac_at_sustained_clock`
                    r["peak"] = r["units"] * PEAK_CLOCK_GHZ / r["cycles_each"]                       # G instructions / s at the part's 2.4 GHz
                    r["frac"] = r["achieved"] / r["peak"]
                    r["peak_at_sustained_clock"] = r["units"] * clock / r["cycles_each"]             # ... at the sustained clock
                    r["frac_at_sustained_clock"] = r["achieved"] / r["peak_at_sustained_clock"]
                    r["roof_ms"] = r["per_wave_step"] * wave_steps * r["cycles_each"] / (r["units"] * clock * 1e9) * 1e3
                bound = max(roofs, key=lambda k: roofs[k]["frac"])   # (the ranking is the same at either clock)
                # Empirical issue model (NOT a roof): in the probes an f16 32x32x16 MFMA keeps the SIMD's vector issue port for ~12-17 cycles
                # (its 24 source + 16 destination registers), not for one 4-cycle slot -- profiles/r02_overlap2_probe.txt, r02_mlp_probe.txt
                model_cycles = (port_cycles - 4.0 * cnt["mfma"]) + cnt["mfma"] * (MFMA_PORT_CYCLES if args.precision != "fp32" else MFMA_F32_CYCLES)
                model_ms = model_cycles * wave_steps / (N_SIMDS * clock * 1e9) * 1e3
                line["roofline"] = {
                    "bound": bound, "achieved": roofs[bound]["achieved"], "peak": roofs[bound]["peak"],
                    "unit": "G wave-gathers/s" if bound == "l1-gather-issue" else "G wave-instructions/s",
                    "frac": roofs[bound]["frac"], "traffic": traffic_now, "traffic_source": traffic_source,
                    "frac_at_sustained_clock": roofs[bound]["frac_at_sustained_clock"], "peak_at_sustained_clock": roofs[bound]["peak_at_sustained_clock"],
                    "issue_cycles_per_wave_instruction": roofs["simd-issue"]["cycles_each"], "peak_clock_ghz": PEAK_CLOCK_GHZ,
                    "kernel": cnt["kernel"], "kernel_ms": k_med, "sustained_clock_ghz": clock,
                    "instructions_per_wave_step": {k: cnt[k] for k in ("valu", "mfma", "vmem_load", "lds", "packed_f32") if k in cnt} | {"gather": n_gather},
                    "roofs": roofs,
                    "empirical_issue_model": {"ms": model_ms, "frac": model_ms / k_med, "priced_at": "sustained clock",
                                              "what": "VALU at their class prices + MFMA x %g cycles of vector-issue-port time per wave-step at the sustained clock "
                                                      "(measured port cost of an MFMA; the f32-input MFMA holds the port for its whole 64 cycles)"
                                                      % (MFMA_PORT_CYCLES if args.precision != "fp32" else MFMA_F32_CYCLES)},
                    "simd_issue_frac": roofs["simd-issue"]["frac"], "l1_gather_issue_frac": roofs["l1-gather-issue"]["frac"], "matrix_pipe_frac": roofs["matrix-pipe"]["frac"],
                    "repriced_r06": "r01-r05 priced every VALU at 4 cycles (simd-issue 0.58); the probe shows 2 / 4 / 8-cycle classes: simd-issue is "
                                    "lower, the L1 gather path (16 cycles per 64-lane gather per CU) is the largest fraction",
                    "note": "bound = the hardware resource with the largest busy fraction.  `frac` prices it at the PART's 2.4 GHz peak clock, "
                            "`frac_at_sustained_clock` at the clock the chip sustains under this kernel (measured in this run; package power limit): the "
                            "difference between the two is power, not scheduling.  The matrix pipe hides plain VALU issued beside it only in part "
                            "(profiles/r02_overlap2_probe.txt, r02_mlp_probe.txt), so the simd-issue and matrix-pipe fractions sum to more "
                            "than 1 and neither reaches it."}
            else:
                line["roofline"] = {"bound": "simd-issue", "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None,
                                    "traffic": traffic_now, "traffic_source": traffic_source, "error": cnt.get("error", "clock probe failed")}
            # Secondary view (north_star: "MFMA utilisation against gfx950 peak"): matrix-core instructions issued per launch are
            # fixed by the kernel (per wave-step of 64 samples: 120 v_mfma_f32_32x32x16_f16 in split precision, 320
            # v_mfma_f32_32x32x2_f32 in exact fp32; rocprofv3 SQ_INSTS_MFMA agrees, profiles/) -- issued flops / kernel time
            # against the dense peak of that MFMA type (MI355X_MICROARCH.md: 2.5 PFLOP/s f16, 157.3 TFLOP/s f32-input).
            mfma_per_step, flop, peak = (120, 2 * 32 * 32 * 16, 2500.0) if args.precision == "fp16x2" else (320, 2 * 32 * 32 * 2, 157.3)
            issued = (W * H * S / 64) * mfma_per_step * flop
            line["roofline_mfma"] = {"bound": "mfma", "achieved": issued / (k_med * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                     "frac": issued / (k_med * 1e-3) / 1e12 / peak,
                                     "frac_at_sustained_clock": (issued / (k_med * 1e-3) / 1e12 / (peak * clock / PEAK_CLOCK_GHZ)) if clock == clock else None,
                                     "algorithmic_tflops": W * H * S * 22784 / (k_med * 1e-3) / 1e12,
                                     "note": "issued MFMA flops (3-term fp16 split, 32-row tile padding) = matrix-pipe busy fraction at the 2.4 GHz "
                                             "peak clock; algorithmic = 22 784 FLOP per sample (SURVEY 8(d))"}
        else:
            # Proposal path: the vector issue port over the whole render call (K2's two marching loops + K1 in bins mode), static
            # instruction counts of the loaded library x the steps each loop runs.  K2's resampling passes (~18 % of its VALU in the PMC
            # profile) are not counted, so the fraction is a lower bound of the port's busy share.
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import kernel_counts

                k2 = kernel_counts.mfma_loops("sn_proposal_kernelILi0ELi5ELi4E")
                k1 = kernel_counts.loop_counts("sn_render_main_kernelILi1ELi%dELi0ELi11E" % (0 if args.precision == "fp32" else 1))
                steps_k2 = list(cfg.num_proposal_samples_per_ray[:cfg.num_proposal_iterations])
                if len(k2) != len(steps_k2):
                    raise RuntimeError("expected one marching loop per proposal net, found %d" % len(k2))
                clock = sustained_clock_ghz(6)
                tiles = ((W + 7) // 8) * ((H + 7) // 8)
                slots = sum(n * (c["valu"] + c["mfma"]) for n, c in zip(steps_k2, k2)) + S * (k1["valu"] + k1["mfma"])
                # (K2's loops: no per-opcode histogram from mfma_loops -- priced at K1's mean cycles per instruction, the same opcode families)
                k1_cycles, _ = issue_port_cycles(k1.get("opcodes") or {"v_unknown": k1["valu"], "v_mfma": k1["mfma"]})
                mean_cycles = k1_cycles / (k1["valu"] + k1["mfma"])
                achieved = slots * tiles / (k_med * 1e-3) / 1e9
                peak = N_SIMDS * PEAK_CLOCK_GHZ / mean_cycles
                line["roofline"] = {
                    "bound": "simd-issue", "achieved": achieved, "peak": peak, "unit": "G wave-instructions/s", "frac": achieved / peak,
                    "frac_at_sustained_clock": achieved / (N_SIMDS * clock / mean_cycles),
                    "issue_cycles_per_wave_instruction": mean_cycles, "peak_clock_ghz": PEAK_CLOCK_GHZ,
                    "traffic": None, "traffic_source": "not measured for this workload", "kernel": "sn_proposal_kernel<0,5,4> + sn_render_main_kernel<1,*> (whole render call)", "kernel_ms": k_med,
                    "sustained_clock_ghz": clock,
                    "instructions_per_wave_step": {"K2 net %d (%d steps)" % (i, n): {k: c.get(k, 0) for k in ("valu", "mfma", "gather")}
                                                   for i, (n, c) in enumerate(zip(steps_k2, k2))} |
                                                  {"K1 bins mode (%d steps)" % S: {k: k1.get(k, 0) for k in ("valu", "mfma", "gather")}},
                    "note": "VALU + MFMA instructions through the one vector issue port of a SIMD (4 cycles each); `frac` at the part's 2.4 GHz, "
                            "`frac_at_sustained_clock` at the clock the chip sustains under this call; static counts (paths the workload does not take included, K2's resampling passes "
                            "excluded).  rocprofv3 PMC of K2 alone: VALU port 0.94 busy (profiles/r02_K2_960x540_pmc_summary.txt)."}
            except Exception as e:  # noqa: BLE001
                line["roofline"] = {"bound": "simd-issue", "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None,
                                    "traffic": None, "error": repr(e)}
        if line["roofline"].get("frac") and args.frames_in_flight > 1 and world == 1:
            # the same roof at the job's frame period (frames overlapped) instead of per launch
            line["roofline"]["frac_at_frame_period"] = line["roofline"]["frac"] * k_med / (elapsed / args.steps * 1e3)
        if not args.no_alt_precision:
            other = "fp32" if args.precision == "fp16x2" else "fp16x2"
            ms = kernel_ms_of(other)
            line["alt_precision"] = {"precision": other, "kernel_ms": ms, "ray_samples_per_s_per_gpu": W * H * S / (ms * 1e-3),
                                     "roofline_hbm_frac": (W * H * bytes_per_ray) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        if world == 1 and not args.no_others and args.workload == "sheet64":
            line["others"] = other_configs(args.precision)
        if world == 1 and not args.no_cpu_baseline and args.workload == "sheet64":
            line["cpu_baseline"] = cpu_baseline(cfg, sd, W, H, S, args.cpu_runs, args.cpu_runs_config4)
        # r05: the driver's record keeps `config`, `roofline` and `cpu_baseline` whole and only the NAMES of every other
        # key -- so the figures a reader needs beside `value` are repeated, compactly, inside `roofline`
        rf = line["roofline"]
        rf["one_launch_ms"] = k_med
        rf["one_launch_ray_samples_per_s"] = W * H * S / (k_med * 1e-3)
        rf["frames_in_flight"] = max(1, args.frames_in_flight)
        rf["value_is"] = ("whole-job rate over the timed steps with %d frames in flight on alternating streams; one launch at a time: one_launch_*"
                          % max(1, args.frames_in_flight))
        if inrun and inrun.get("l2_hit_rate") is not None:
            rf["l2_hit_rate"] = inrun["l2_hit_rate"]
        if traffic_now:
            rf["traffic_over_algorithmic"] = traffic_now / (W * H * bytes_per_ray)
            rf["traffic_frac_of_hbm_peak"] = traffic_now / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBPS
        alt = line.get("alt_precision")
        if alt:
            key = "exact_fp32" if alt["precision"] == "fp32" else "split_fp16x2"
            rf[key] = {"kernel_ms": alt["kernel_ms"], "ray_samples_per_s": alt["ray_samples_per_s_per_gpu"],
                       "what": "the same frame in the other MFMA arithmetic, one launch at a time, after the timed region (HIP events, mean of 10)"}
        if "others" in line:
            oc = {}
            for leg in line["others"]:
                c = leg.get("config", "")
                if "error" in leg:
                    oc.setdefault("errors", []).append({"config": c, "error": leg["error"]})
                elif c.startswith("BASELINE.json configs[3]"):
                    oc["configs3"] = {"what": "1920x1080, proposal nets 256 + 96 + 48 main samples, random-weight scene", "ms_per_frame": leg["ms_per_frame"],
                                      "kernel_ms_per_launch": leg["kernel_ms_per_launch"], "field_evaluations_per_s": leg["field_evaluations_per_s"],
                                      "frac": (leg.get("roofline") or {}).get("frac"), "frac_at_sustained_clock": (leg.get("roofline") or {}).get("frac_at_sustained_clock")}
                elif c.startswith("BASELINE.json configs[4]"):
                    oc["configs4"] = {"what": "DatasetGenerator.generate_dataset, 8 reference + 50 views at 800x800, PNG writes off, 1 GPU", "total_ms": leg["total_ms"],
                                      "ms_per_view": leg["ms_per_view"], "render_ms_per_view": leg.get("render_ms_per_view")}
                elif c.startswith("trained scene"):
                    oc["trained"] = {"scene_fit_s": (leg.get("scene") or {}).get("seconds"),
                                     "legs": [{"frame": x["frame"], "ms_early_term_off": x["ms_per_frame"]["early_term_off"], "ms_early_term_on": x["ms_per_frame"]["early_term_on"],
                                               "speedup": x["ms_per_frame"]["speedup"], "bit_identical": x["bit_identical_on_vs_off"],
                                               "skipped_wave_step_fraction": {k: v["skipped_fraction"] for k, v in x["wave_steps"].items()},
                                               "accumulation_above_0.99": x["picture"]["accumulation_above_0.99"]} for x in leg.get("legs", [])]}
                elif c.startswith("BASELINE configs[1] frame with T = 2^"):
                    oc["T21"] = {k: leg.get(k) for k in ("ms_per_frame", "kernel_ms_per_launch", "traffic_bytes_per_launch", "traffic_over_algorithmic",
                                                         "traffic_frac_of_hbm_peak", "l2_hit_rate", "simd_issue_frac", "binding_roof")}
            rf["other_configs"] = oc
        line["roofline"] = flat_roofline(rf, line)   # the leading keys: flat scalars, in the order the driver's record keeps
        if world > 1:
            spread, tiles_check = run_post_checks(line)
            if spread:
                line["config"].update({"rank_kernel_ms_min": spread["min"], "rank_kernel_ms_max": spread["max"], "rank_kernel_ms_mean": spread["mean"],
                                       "slowest_rank": spread["slowest_rank"]})
            if tiles_check:
                line["config"]["gathered_tiles_bit_identical"] = (tiles_check.get("mismatching_tiles_all_ranks") == 0 if "error" not in tiles_check
                                                                  else tiles_check["error"])
            line["rank_kernel_ms"], line["gathered_tiles_check"] = spread, tiles_check
            gather_by_strategy, gather_err = gather_diagnostics(line)
            gather_ms = gather_by_strategy.get(args.gather_strategy)
            line["gather_ms"] = {"exposed": None, "error": gather_err, "exposed_by_strategy": gather_by_strategy} if gather_ms is None else {
                "exposed": gather_ms, "strategy": args.gather_strategy, "exposed_by_strategy": gather_by_strategy,
                "strategies": "all_gather = RCCL all_gather_into_tensor / gather (ring or tree, RCCL's choice); p2p = world - 1 direct isend / irecv "
                              "pairs per rank (one per xGMI link); all_to_all = the same pushes as one all_to_all_single (signerf_amd/sheet.py)",
                "in_timed_region": "overlapped with the next step's renders (depth-1 pipeline); every gather completes inside it",
                "what": "HIP-event time on the caller's stream of issuing the tile gather and waiting for it with nothing to overlap "
                        "(median of 5, after a barrier), i.e. what each step would pay without the pipeline",
                "bytes_sent_per_rank": (len(mine) if strong else 1) * H * W * 16}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
