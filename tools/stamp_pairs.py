#!/usr/bin/env python3
"""How the phases of a CU's two workgroups line up (`make stamppairs`: libneddf_hip_stamppairs.so, NEDDF_STAMP_FILE).

    NEDDF_LIB_PATH=neddf_amd/csrc/libneddf_hip_stamppairs.so NEDDF_STAMP_FILE=/tmp/sp.bin NEDDF_PROBE_DTYPE=f16_split python tools/pmc_probe.py
    python tools/stamp_pairs.py /tmp/sp.bin [n_layers=7]

Workgroups b and b + 256 (b = 0..3) record four consecutive tiles each (wave 0's stamps; the workgroup's waves move together between
barriers).  Every interval between two stamps is classified as M (a matrix-product phase), V (an epilogue / multiply-and-store phase:
vector work) or O (barriers, encoding, heads, tail); for every pair on the same CU the overlap matrix of the two timelines is printed."""
import sys

import numpy as np

SLOTS, TILES = 160, 4
path = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 7
raw = np.fromfile(path, dtype=np.uint64).reshape(8, 8, SLOTS * TILES)
kinds = ["O", "O"]                                  # encode, bar
for l in range(L):
    kinds += ["M", "O", "V", "O"]
kinds += ["O", "O", "O", "O"]                       # heads+handoff, bar, gL store, bar
for l in range(L - 1):
    kinds += ["O", "M", "O", "V", "O"]
kinds += ["O", "O"]                                 # tail, next-tile bar
per_tile = len(kinds) + 1


def timeline(b):
    st = raw[b, 0, 1:1 + per_tile * TILES].astype(np.int64)
    hw = int(raw[b, 0, 0])
    segs = []
    for t in range(TILES):
        s = st[t * per_tile:(t + 1) * per_tile]
        if (s == 0).any():
            break
        for i, k in enumerate(kinds):
            segs.append((s[i], s[i + 1], k))
    return hw, segs


def cu_of(hw):
    lo, xcc = hw & 0xffffffff, hw >> 32
    return (xcc & 7, (lo >> 13) & 7, (lo >> 12) & 1, (lo >> 8) & 15)


for b in range(4):
    hwa, A = timeline(b)
    hwb, B = timeline(b + 4)
    same = cu_of(hwa) == cu_of(hwb)
    print("workgroups %d and %d: CU %s / %s -> %s" % (b, b + 256, cu_of(hwa), cu_of(hwb), "SAME CU" if same else "different CUs"))
    if not A or not B:
        print("  (no stamps)")
        continue
    lo, hi = max(A[0][0], B[0][0]), min(A[-1][1], B[-1][1])
    if hi <= lo:
        print("  (the stamped windows do not overlap in time)")
        continue
    mat = {}
    ia = ib = 0
    t = lo
    while t < hi:
        while ia < len(A) and A[ia][1] <= t:
            ia += 1
        while ib < len(B) and B[ib][1] <= t:
            ib += 1
        if ia >= len(A) or ib >= len(B):
            break
        nxt = min(A[ia][1], B[ib][1], hi)
        if A[ia][0] <= t and B[ib][0] <= t:
            mat[(A[ia][2], B[ib][2])] = mat.get((A[ia][2], B[ib][2]), 0) + (nxt - t)
        t = nxt
    tot = float(sum(mat.values()))
    print("  common window %.0f k cycles; share of it by (phase of %d, phase of %d):  M = product, V = epilogue, O = other" % (tot / 1e3, b, b + 256))
    for ka in "MVO":
        print("   ", "  ".join("%s%s %5.1f%%" % (ka, kb, 100 * mat.get((ka, kb), 0) / tot) for kb in "MVO"))
    pa = {k: sum(v for (x, _), v in mat.items() if x == k) / tot for k in "MVO"}
    pb = {k: sum(v for (_, y), v in mat.items() if y == k) / tot for k in "MVO"}
    print("    if independent: MM %.1f%%  VV %.1f%%  MV+VM %.1f%%" % (100 * pa["M"] * pb["M"], 100 * pa["V"] * pb["V"], 100 * (pa["M"] * pb["V"] + pa["V"] * pb["M"])))
