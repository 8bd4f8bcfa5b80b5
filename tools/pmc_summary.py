#!/usr/bin/env python3
"""Summarise tools/pmc_passes.sh output: per-kernel average of every counter over the dispatches of `--kernel` (default
sn_render_main).  Writes a text table to stdout."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "sn_render_main"
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        sys.exit("no counters for kernel pattern %r under %s" % (pat, d))
    print("kernel pattern: %s   (values averaged over dispatches, summed over XCDs/SEs as rocprofv3 reports them)" % pat)
    for k in sorted(acc):
        v = acc[k]
        print("%-40s n=%3d  avg %18.1f" % (k, len(v), sum(v) / len(v)))
    a = {k: sum(v) / len(v) for k, v in acc.items()}
    print()
    if "FETCH_SIZE" in a:
        print("FETCH_SIZE (KiB as reported) -> %.3f GB per launch as reported; x2 (gfx950 128-B-request correction for wide streams, "
              "uncalibrated for 8-B gathers) -> %.3f GB" % (a["FETCH_SIZE"] * 1024 / 1e9, 2 * a["FETCH_SIZE"] * 1024 / 1e9))
    if "WRITE_SIZE" in a:
        print("WRITE_SIZE -> %.4f GB per launch" % (a["WRITE_SIZE"] * 1024 / 1e9))
    if all(k in a for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        n32, n64, n128, n = a["TCC_EA0_RDREQ_32B_sum"], a["TCC_EA0_RDREQ_64B_sum"], a["TCC_EA0_RDREQ_128B_sum"], a["TCC_EA0_RDREQ_sum"]
        other = n - n32 - n64 - n128
        print("EA read requests: total %.3e = 32B %.3e + 64B %.3e + 128B %.3e (+ other %.3e) -> bytes by size classes %.3f GB" %
              (n, n32, n64, n128, other, (32 * n32 + 64 * n64 + 128 * n128) / 1e9))
    if "TCC_HIT_sum" in a and "TCC_MISS_sum" in a:
        print("L2 hit rate %.1f %%" % (100 * a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
        print("MFMA busy / SQ busy cycles = %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"]))
    if "SQ_WAVE_CYCLES" in a:
        for k in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
            if k in a:
                print("%s / SQ_WAVE_CYCLES = %.3f" % (k, a[k] / a["SQ_WAVE_CYCLES"]))


    if "--json" in sys.argv:  # python tools/pmc_summary.py DIR PATTERN --json PRECISION  -> updates profiles/traffic.json
        import json
        prec = sys.argv[sys.argv.index("--json") + 1]
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
        try:
            cur = json.load(open(path))
        except (OSError, ValueError):
            cur = {}
        rd = 32 * a.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * a.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * a.get("TCC_EA0_RDREQ_128B_sum", 0)
        wr = a.get("WRITE_SIZE", 0) * 1024
        cur[prec] = {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "commit": os.environ.get("SN_COMMIT", "unknown"),
                     "source": "tools/pmc_passes.sh: TCC_EA0_RDREQ_{32,64,128}B_sum x size + WRITE_SIZE KiB, kernel " + pat}
        json.dump(cur, open(path, "w"), indent=1)
        print("updated", path)


if __name__ == "__main__":
    main()
