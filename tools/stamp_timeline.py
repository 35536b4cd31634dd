#!/usr/bin/env python3
"""Per-phase cycle table of the reverse-mode distance kernel from a `make stamp` build (libneddf_hip_stamp.so, NEDDF_STAMP_FILE).

    NEDDF_LIB_PATH=neddf_amd/csrc/libneddf_hip_stamp.so NEDDF_STAMP_FILE=/tmp/st.bin NEDDF_PROBE_DTYPE=bf16 python tools/pmc_probe.py
    python tools/stamp_timeline.py /tmp/st.bin [n_layers=7]

Stamp order inside one tile (field_kernels.hip ddf_rev_kernel): 0 tile start, 1 encoding done, 2 barrier; per forward layer l:
product done, barrier, epilogue done, barrier; then heads/hand-off done, barrier, g_L stored, barrier; per reverse layer
(n_layers - 1 of them): setup done, product done, barrier (y' requested before it), multiply+store done, barrier; tail done; tile end."""
import sys

import numpy as np

BLOCKS, WAVES, SLOTS = 8, 8, 160
path = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 7
# (`make stamp`: wave w of workgroup b writes at (8 b + w) * SLOTS; the buffer is allocated four times that size for the pair mode of
# `make stamppairs`, tools/stamp_pairs.py)
raw = np.fromfile(path, dtype=np.uint64)[:BLOCKS * WAVES * SLOTS].reshape(BLOCKS, WAVES, SLOTS)[:, :4, :].astype(np.int64)
names = ["encode", "bar"]
for l in range(L):
    names += ["F%d product" % l, "F%d bar" % l, "F%d epilogue" % l, "F%d bar" % l]
names += ["heads+handoff", "bar", "gL store", "bar"]
for l in range(L - 1, 0, -1):
    names += ["R%d setup/skip" % l, "R%d product" % l, "R%d bar(+y' req)" % l, "R%d y' mul+store" % l, "R%d bar" % l]
names += ["tail"]
# NEDDF_FUSED=1: one more stamp after the colour trunk that runs on the same tile
fused = bool((raw[:, :, len(names) + 2] != 0).all())
if fused:
    names += ["colour trunk (fused)"]
names += ["next-tile bar"]
n = len(names) + 1
ok = raw[:, :, :n]
if (ok[:, :, 1:] == 0).any():
    print("warning: some stamps are missing (kernel took another path?)", int((ok == 0).sum()))
d = np.diff(ok, axis=2).astype(np.float64)          # [block][wave][phase]
tot = (ok[:, :, -1] - ok[:, :, 0]).astype(np.float64)
print("tile span per wave (cycles): mean %.0f  min %.0f  max %.0f   (%d workgroups x 4 waves)" % (tot.mean(), tot.min(), tot.max(), BLOCKS))
# the constant 100 MHz clock at the start and the end of the stamped tile (last two slots): the shader clock the part held
wall = (raw[:, :, SLOTS - 1] - raw[:, :, SLOTS - 2]).astype(np.float64)
if (wall > 0).all():
    ghz = tot / (wall / 100e6) / 1e9
    print("shader clock while the tile ran (cycles of the tile / its time on the constant 100 MHz clock): mean %.3f GHz  min %.3f  max %.3f" % (ghz.mean(), ghz.min(), ghz.max()))
print("%-20s %10s %10s %10s %7s" % ("phase", "mean", "min", "max", "share"))
groups = {}
for i, nm in enumerate(names):
    v = d[:, :, i]
    print("%-20s %10.0f %10.0f %10.0f %6.1f%%" % (nm, v.mean(), v.min(), v.max(), 100 * v.mean() / tot.mean()))
    key = ("barrier wait" if "bar" in nm else "forward product" if nm.startswith("F") and "product" in nm else
           "forward epilogue" if nm.startswith("F") else "reverse product" if nm.startswith("R") and "product" in nm else
           "reverse y' mul+store" if nm.startswith("R") and "mul" in nm else "reverse setup/skip share" if nm.startswith("R") else nm)
    groups[key] = groups.get(key, 0.0) + v.mean()
print()
for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
    print("%-28s %10.0f cycles %6.1f%%" % (k, v, 100 * v / tot.mean()))
