"""Host-side pieces of bench.py that need no GPU: the watchdog around the N > 1 diagnostic legs and the cgroup CPU quota."""
import threading
import time

import bench


def test_watchdog_fires_only_on_a_call_that_does_not_return():
    fired = threading.Event()
    assert bench.run_with_watchdog(lambda: 7, 0.05, fired.set) == 7
    time.sleep(0.15)
    assert not fired.is_set()                       # returned in time: never
    release = threading.Event()

    def stuck():                                    # stands for a collective that hangs; the real on_timeout ends the process
        release.wait(5.0)
        return "late"

    t0 = time.perf_counter()
    out = bench.run_with_watchdog(stuck, 0.1, lambda: (fired.set(), release.set()))
    assert fired.is_set() and out == "late" and time.perf_counter() - t0 < 2.0


def test_watchdog_lets_exceptions_through_and_disarms():
    fired = threading.Event()
    try:
        bench.run_with_watchdog(lambda: 1 / 0, 0.05, fired.set)
    except ZeroDivisionError:
        pass
    else:
        raise AssertionError("the exception was swallowed")
    time.sleep(0.15)
    assert not fired.is_set()


def test_cpu_quota_is_none_or_positive():
    q = bench.cpu_quota()
    assert q is None or q > 0
    cores, logical = bench.physical_cores()
    assert 1 <= cores <= logical


def test_roofline_leads_with_flat_scalars_in_the_order_the_driver_keeps():
    """The driver's record keeps the leading scalar keys of `roofline` and drops nested dicts (VERDICT r05 "weak 4"): the figures that
    must survive -- one-launch and exact-fp32 times, the HBM line as a RATIO, measured traffic, cache hit rate, the other configurations --
    are the first keys, flat, whatever order the run computed them in."""
    rf = {"note": "x" * 300, "roofs": {"simd-issue": {}}, "bound": "simd-issue", "achieved": 357.8, "peak": 614.4, "unit": "G wave-instructions/s", "frac": 0.58,
          "traffic": 5.4e9, "kernel_ms": 2.78, "sustained_clock_ghz": 1.75, "frac_at_sustained_clock": 0.79, "issue_cycles_per_wave_instruction": 3.2, "simd_issue_frac": 0.46, "l1_gather_issue_frac": 0.5,
          "one_launch_ray_samples_per_s": 14.7e9, "frames_in_flight": 2, "l2_hit_rate": 0.62, "other_configs": {"configs3": {}}}
    line = {"dtype": "fp16x2-split multiply ...", "roofline": rf,
            "alt_precision": {"precision": "fp32", "kernel_ms": 7.0, "ray_samples_per_s_per_gpu": 5.85e9},
            "roofline_hbm": {"frac": 1.88, "traffic_over_algorithmic": 0.13, "traffic_frac_of_hbm_peak": 0.24},
            "roofline_mfma": {"frac": 0.36},
            "others": [{"config": "BASELINE.json configs[3]: 1920x1080 ...", "ms_per_frame": 15.1},
                       {"config": "BASELINE.json configs[4]: DatasetGenerator.generate_dataset, ...", "ms_per_view": 5.0},
                       {"config": "tiny-cuda-nn grid ...", "error": "boom"},
                       {"config": "trained scene (tools/make_trained_scene.py): ...",
                        "legs": [{"frame": "800x800, 256 + 96 + 48 samples", "ms_per_frame": {"early_term_on": 4.17, "early_term_off": 5.06}},
                                 {"frame": "800x800, 64 uniform samples, no proposal nets", "ms_per_frame": {"early_term_on": 2.22}}]},
                       {"config": "BASELINE configs[1] frame with T = 2^21 rows per level (256 MiB table)", "kernel_ms_per_launch": 2.8}]}
    out = bench.flat_roofline(rf, line)
    keys = tuple(out)
    assert keys[:len(bench.ROOFLINE_LEADING_KEYS)] == bench.ROOFLINE_LEADING_KEYS
    assert keys[:6] == ("bound", "achieved", "peak", "unit", "frac", "traffic")          # the contract's own keys first
    lead = keys[:21]                                                                     # what r05's record kept: 21 scalars
    for k in ("kernel_ms", "one_launch_ms", "one_launch_ray_samples_per_s", "exact_fp32_ms", "exact_fp32_ray_samples_per_s", "hbm_algorithmic_ratio",
              "traffic_over_algorithmic", "traffic_frac_of_hbm_peak", "l2_hit_rate", "mfma_frac", "sustained_clock_ghz", "configs3_ms_per_frame",
              "configs4_ms_per_view", "trained_800_ms", "T21_ms"):
        assert k in lead, k
    assert all(out[k] is None or isinstance(out[k], (int, float, str)) for k in bench.ROOFLINE_LEADING_KEYS)
    assert all(len(out[k]) <= 128 for k in bench.ROOFLINE_LEADING_KEYS if isinstance(out[k], str))
    assert out["one_launch_ms"] == 2.78 and out["exact_fp32_ms"] == 7.0 and out["exact_fp32_ray_samples_per_s"] == 5.85e9
    assert out["hbm_algorithmic_ratio"] == 1.88 and out["traffic_over_algorithmic"] == 0.13 and out["mfma_frac"] == 0.36
    assert out["configs3_ms_per_frame"] == 15.1 and out["configs4_ms_per_view"] == 5.0 and out["trained_800_ms"] == 4.17 and out["T21_ms"] == 2.8
    assert out["roofs"] == {"simd-issue": {}} and out["note"] == rf["note"] and "frac_at_peak_clock" not in out   # the rest follows, nothing lost
    # an exact-fp32 run is its own fp32 figure; missing legs give None, not a KeyError
    out32 = bench.flat_roofline({"kernel_ms": 7.0, "one_launch_ray_samples_per_s": 5.85e9}, {"dtype": "f32 (exact fp32 MFMA)"})
    assert out32["exact_fp32_ms"] == 7.0 and out32["exact_fp32_ray_samples_per_s"] == 5.85e9 and out32["configs3_ms_per_frame"] is None


def test_issue_port_pricing_by_opcode_class():
    """r06: the vector issue port's price of a wave64 instruction by opcode class (profiles/r06_valu_issue_probe.txt): 2 cycles for the plain
    fp32 / integer ALU ops, 8 for transcendentals and the half-wave swap, 20 for a VCC-masked select, 4 for everything else and an MFMA's issue."""
    cyc, by = bench.issue_port_cycles({"v_fmac_f32": 10, "v_cvt_pkrtz_f16_f32": 5, "v_exp_f32": 2, "v_cndmask_b32": 1, "v_mfma_f32_32x32x16_f16": 3, "v_never_heard_of": 1})
    assert cyc == 10 * 2 + 5 * 4 + 2 * 8 + 20 + 3 * 4 + 4
    assert by == {"2-cycle": 10, "4-cycle": 6, "8-cycle": 2, "20-cycle": 1, "mfma issue (4)": 3}
