#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -S --offload-device-only field_kernels.hip -o fk.s
    python tools/isa_mix.py fk.s ddf_rev_kernelILi2ENS_9OpsBF16RTILi256EEELb0 [top]
"""
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ops = Counter()
for l in lines[start:end]:
    if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        ops[l.split()[0]] += 1
print(lines[start], sum(ops.values()), "instructions")
groups = Counter()
for k, v in ops.items():
    if k.startswith("v_mfma"): groups["mfma"] += v
    elif k in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32"): groups["transcendental"] += v
    elif k.startswith("v_pk_"): groups["valu packed"] += v
    elif k.startswith("v_"): groups["valu"] += v
    elif k.startswith("ds_"): groups["lds"] += v
    elif k.startswith(("buffer_", "global_", "scratch_", "flat_")): groups["vmem"] += v
    elif k.startswith("s_"): groups["scalar"] += v
    else: groups["other"] += v
for k, v in groups.most_common(): print(f"  {k:16s} {v}")
for k, v in ops.most_common(top): print(f"    {k:28s} {v}")
