#!/usr/bin/env python3
"""Where a kernel's register spills sit: scratch loads/stores and MFMAs per basic block of one kernel of an assembly listing
(hipcc -S --cuda-device-only).   python tools/isa_spills.py /tmp/fk.s 'ddf_rev_kernelILi2ELi4ELi2ENS_7OpsF32TILi256EEELb0ELb1'"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(_Z\S*' + re.escape(pat) + r'\S*): .*?\n(.*?)s_endpgm', txt, re.S | re.M)
if not m:
    sys.exit("kernel not found")
body = m.group(2).split('\n')
lab, sc, mf, n = "entry", Counter(), Counter(), Counter()
order = []
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        lab = l.split(':')[0]
        order.append(lab)
    if re.match(r'^\s+[a-z]', l):
        n[lab] += 1
    if 'scratch_' in l:
        sc[lab] += 1
    if 'v_mfma' in l:
        mf[lab] += 1
print(m.group(1)[:120])
print("instructions %d, scratch ops %d, mfma %d" % (sum(n.values()), sum(sc.values()), sum(mf.values())))
for lab in ["entry"] + order:
    if sc[lab] or mf[lab]:
        print("%-12s instr %5d  mfma %4d  scratch %3d" % (lab, n[lab], mf[lab], sc[lab]))
