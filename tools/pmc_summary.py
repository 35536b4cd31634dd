#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files found under a directory.

    python tools/pmc_summary.py gpurun_out/pmc > profiles/xxx_pmc.csv
"""
import collections
import csv
import glob
import os
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for k in agg:
    if "neddf" not in k:
        continue
    for c, v in sorted(agg[k].items()):
        w.writerow([k, c, len(v), "%.1f" % (sum(v) / len(v))])
