#!/usr/bin/env python3
"""Dump the per-kernel stats of a rocprofv3 run (sqlite .db or *_kernel_stats.csv) as a text table.

    python tools/rocprof_summary.py gpurun_out/prof_dir > profiles/r01_xxx_kernel_stats.txt
"""
import csv
import glob
import os
import sqlite3
import sys


def main(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    csvs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    print("%-70s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    if csvs:
        for r in csv.DictReader(open(csvs[0])):
            print("%-70s %8s %14.1f %12.1f %8.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                      float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    elif dbs:
        con = sqlite3.connect(dbs[0])
        for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%-70s %8d %14.1f %12.1f %8.2f" % (name[:70], calls, total, avg, pct))
        print("\nper-dispatch resources of the sn_* kernels:")
        q = ("select name, min(vgpr_count), min(accum_vgpr_count), min(sgpr_count), min(lds_size), min(scratch_size), min(grid_x), min(workgroup_x) "
             "from kernels where name like '%sn_%' group by name")
        for row in con.execute(q):
            print("  %-60s vgpr %s agpr %s sgpr %s lds %s scratch %s grid %s wg %s" % ((row[0][:60],) + row[1:]))
    else:
        sys.exit("no rocprofv3 output found under " + d)


if __name__ == "__main__":
    main(sys.argv[1])
