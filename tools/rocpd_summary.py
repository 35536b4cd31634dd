#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2's default --kernel-trace --stats
output) as CSV: kernel, calls, total_ms, avg_ms, pct (the top_kernels view reports microseconds); plus launch geometry and
register counts of the first dispatch of each kernel.

    python tools/rocpd_summary.py gpurun_out/prof/r1_results.db > profiles/xxx.csv
"""
import csv
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    geo = {}
    for r in c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size "
                       "from kernels group by name"):
        geo[r[0]] = r[1:]
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_ms", "avg_ms", "pct", "grid_x", "workgroup_x", "lds_bytes", "vgpr", "agpr", "sgpr",
                "scratch"])
    for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        short = name if len(name) < 120 else name[:117] + "..."
        w.writerow([short, calls, "%.1f" % (total / 1e3), "%.2f" % (avg / 1e3), "%.3f" % pct] + list(geo.get(name, [""] * 7)))


if __name__ == "__main__":
    main(sys.argv[1])
