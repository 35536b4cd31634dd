#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this box: known byte counts (tools/traffic_calib.hip, its KNOWN lines) against the
counter values of two --pmc passes.  factor = known bytes / (counter x 1024); bench.py applies factor_read / factor_write to the field
kernels' counters (profiles/pmc_traffic.json carries them).

    python tools/traffic_calib.py <dir with fetch/ write/ subdirs of rocprofv3 csv output> <stdout of traffic_calib> [out.json]"""
import csv
import glob
import json
import os
import re
import sys

d, log = sys.argv[1], sys.argv[2]
known = {}
for ln in open(log):
    m = re.match(r"KNOWN (\S+) read (\d+) write (\d+)\s+# (.*)", ln)
    if m:
        known[m.group(1)] = dict(read=int(m.group(2)), write=int(m.group(3)), what=m.group(4).strip())
vals = {}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for path in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            key = (name, r.get("Dispatch_Id"))
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        for (name, _), v in per.items():
            vals.setdefault(name, {})[counter] = v
out = {}
print("%-24s %14s %14s %10s %10s   %s" % ("kernel", "FETCH_SIZE KB", "WRITE_SIZE KB", "f_read", "f_write", "what"))
for name, k in known.items():
    v = vals.get(name, {})
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    fr = k["read"] / (f * 1024) if f and k["read"] else None
    fw = k["write"] / (w * 1024) if w and k["write"] else None
    out[name] = dict(known_read=k["read"], known_write=k["write"], FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, factor_read=fr, factor_write=fw, what=k["what"])
    print("%-24s %14s %14s %10s %10s   %s" % (name, "%.0f" % f if f is not None else "-", "%.0f" % w if w is not None else "-",
                                               "%.3f" % fr if fr else "-", "%.3f" % fw if fw else "-", k["what"]))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
