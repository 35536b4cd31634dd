#!/usr/bin/env python3
"""Per-phase cycle table of the colour-trunk kernel from a `make stamp` build (libneddf_hip_stamp.so, NEDDF_STAMP_FILE_COL).

    NEDDF_LIB_PATH=neddf_amd/csrc/libneddf_hip_stamp.so NEDDF_STAMP_FILE_COL=/tmp/sc.bin python tools/pmc_probe.py
    python tools/stamp_timeline_col.py /tmp/sc.bin [n_layers=3] [waves=4]

Stamp order inside one tile (field_kernels.hip col_trunk_kernel): 0 tile start, 1 small-input columns zeroed + barrier, 2 encodings
done, 3 barrier, 4 features requested + small-input product done, 5 barrier, 6 features in LDS, 7 barrier; per layer: product done,
barrier, epilogue done, barrier; then head dot products done, barrier, outputs written, tile end (barrier + next tile index).
Also prints the matrix-pipe floor of the tile (fp32: 64 cycles per v_mfma_f32_32x32x2_f32) so that the share of the tile a wave
spends outside its own products can be read against it."""
import sys

import numpy as np

BLOCKS, WAVES, SLOTS = 8, 8, 160
path = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 4
# (wave w of workgroup b writes at (8 b + w) * SLOTS; the buffer is allocated four times that size for the pair mode)
raw = np.fromfile(path, dtype=np.uint64)[:BLOCKS * WAVES * SLOTS].reshape(BLOCKS, WAVES, SLOTS)[:, :NW, :].astype(np.int64)
names = ["zero small-input columns + bar", "encode pos / dir / normal", "bar", "feature request + small-input product", "bar",
         "features -> LDS", "bar"]
for l in range(L):
    names += ["L%d product" % l, "L%d bar" % l, "L%d epilogue" % l, "L%d bar" % l]
names += ["head dots (W -> 3)", "bar", "outputs", "tile-end bar"]
n = len(names) + 1
ok = raw[:, :, :n]
if (ok[:, :, 1:] == 0).any():
    print("warning: some stamps are missing (kernel took another path?)", int((ok[:, :, 1:] == 0).sum()))
d = np.diff(ok, axis=2).astype(np.float64)          # [block][wave][phase]
tot = (ok[:, :, -1] - ok[:, :, 0]).astype(np.float64)
print("tile span per wave (cycles): mean %.0f  min %.0f  max %.0f   (%d workgroups x %d waves)" % (tot.mean(), tot.min(), tot.max(), BLOCKS, NW))
# the constant 100 MHz clock at the start and the end of the stamped tile (last two slots): the shader clock the part held
wall = (raw[:, :, SLOTS - 1] - raw[:, :, SLOTS - 2]).astype(np.float64)
if (wall > 0).all():
    ghz = tot / (wall / 100e6) / 1e9
    print("shader clock while the tile ran (cycles of the tile / its time on the constant 100 MHz clock): mean %.3f GHz  min %.3f  max %.3f" % (ghz.mean(), ghz.min(), ghz.max()))
print("%-40s %10s %10s %10s %7s" % ("phase", "mean", "min", "max", "share"))
groups = {}
for i, nm in enumerate(names):
    v = d[:, :, i]
    print("%-40s %10.0f %10.0f %10.0f %6.1f%%" % (nm, v.mean(), v.min(), v.max(), 100 * v.mean() / tot.mean()))
    key = ("barrier wait" if nm.endswith("bar") else "layer products" if "product" in nm and nm.startswith("L") else
           "layer epilogues" if "epilogue" in nm else nm)
    groups[key] = groups.get(key, 0.0) + v.mean()
print()
for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
    print("%-40s %10.0f cycles %6.1f%%" % (k, v, 100 * v / tot.mean()))
