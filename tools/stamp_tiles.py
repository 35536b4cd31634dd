#!/usr/bin/env python3
"""Tiles per workgroup of a persistent field kernel from a `make stamp` build: how evenly do the workgroups of one launch progress?

    python tools/stamp_tiles.py <NEDDF_STAMP_FILE or NEDDF_STAMP_FILE_COL dump>

Behind the phase stamps the dump carries, per workgroup: the tiles it took from the queue | XCC_ID << 20 | HW_ID << 24, and its
first / last moment on the constant 100 MHz clock and its shader cycles between the two (the clock the part held over the launch).  With a dynamic tile queue a workgroup's tile count is its speed."""
import sys

import numpy as np

BLOCKS, WAVES, SLOTS, PAIR, TAIL = 8, 8, 160, 4, 4096
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
base = BLOCKS * WAVES * SLOTS * PAIR
w = raw[base:base + TAIL]
t0 = raw[base + TAIL:base + 2 * TAIL].astype(np.int64)
t1 = raw[base + 2 * TAIL:base + 3 * TAIL].astype(np.int64)
cyc = raw[base + 3 * TAIL:base + 4 * TAIL].astype(np.int64) if raw.size >= base + 4 * TAIL else None
used = w != 0
n = int(used.sum())
tiles = (w[used] & np.uint64(0xfffff)).astype(np.int64)
xcc = ((w[used] >> np.uint64(20)) & np.uint64(15)).astype(np.int64)
hw = (w[used] >> np.uint64(24)).astype(np.int64)
cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
t0, t1 = t0[used], t1[used]
start = t0.min()
print("workgroups %d, tiles %d; per workgroup: mean %.1f  min %d  p10 %d  median %d  p90 %d  max %d" %
      (n, tiles.sum(), tiles.mean(), tiles.min(), np.percentile(tiles, 10), np.median(tiles), np.percentile(tiles, 90), tiles.max()))
dur = (t1 - t0) / 100.0
print("lifetime per workgroup (us): mean %.0f  min %.0f  max %.0f; started after the first by at most %.1f us; kernel %.0f us" %
      (dur.mean(), dur.min(), dur.max(), (t0.max() - start) / 100.0, (t1.max() - start) / 100.0))
if cyc is not None and (cyc[used] > 0).all():
    ghz = cyc[used] / (dur * 1e-6) / 1e9        # shader cycles of a workgroup's whole life over its time on the 100 MHz clock
    print("shader clock over the LAUNCH (every workgroup's cycles / its lifetime): mean %.3f GHz  min %.3f  max %.3f; cycles per tile: mean %.0f" %
          (ghz.mean(), ghz.min(), ghz.max(), (cyc[used] / np.maximum(tiles, 1)).mean()))
print("tiles of workgroups 0..7:", tiles[:8].tolist())
idx = np.nonzero(used)[0]
print("per XCD: " + "  ".join("%d: %.1f" % (x, tiles[xcc == x].mean()) for x in sorted(set(xcc.tolist()))))
print("per shader engine (all XCDs): " + "  ".join("%d: %.1f" % (x, tiles[se == x].mean()) for x in sorted(set(se.tolist()))))
print("per CU index within its shader array: " + "  ".join("%d: %.1f" % (x, tiles[cu == x].mean()) for x in sorted(set(cu.tolist()))))
# the two workgroups of a CU
key = xcc * 4096 + se * 256 + sh * 16 + cu
pairs = {}
for k, t, b in zip(key.tolist(), tiles.tolist(), idx.tolist()):
    pairs.setdefault(k, []).append((t, b))
sums = np.array([sum(t for t, _ in v) for v in pairs.values()])
print("CUs %d; workgroups per CU: %s; tiles per CU: mean %.1f  min %d  max %d" %
      (len(pairs), sorted(set(len(v) for v in pairs.values())), sums.mean(), sums.min(), sums.max()))
diffs = [abs(v[0][0] - v[1][0]) for v in pairs.values() if len(v) == 2]
if diffs:
    print("difference between a CU's two workgroups: mean %.1f  max %d; block index distance of the two: %s" %
          (np.mean(diffs), max(diffs), sorted(set(abs(v[0][1] - v[1][1]) for v in pairs.values() if len(v) == 2))[:8]))
hist, edges = np.histogram(tiles, bins=12)
print("histogram of tiles per workgroup:", " ".join("%d-%d:%d" % (edges[i], edges[i + 1], hist[i]) for i in range(len(hist))))
