#!/bin/bash
# Round 5: does a tile take the same time in the middle / at the end of a 2^23-point launch as at its start, and at which clock?
# (stamp builds of tile 6 / 128 / 240 of the first eight workgroups, constant 100 MHz clock beside the cycle counter) + the launch
# times of the stamp build itself.
ROOT=$PWD
O=$ROOT/gpurun_out/r5clock2
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for dt in fp32 bf16; do
  for t in "" 128 240; do
    LIB=$ROOT/neddf_amd/csrc/libneddf_hip_stamp$t.so
    [ -f $LIB ] || continue
    NEDDF_LIB_PATH=$LIB NEDDF_STAMP_FILE=$O/ddf_${dt}_t$t.bin NEDDF_STAMP_FILE_COL=$O/col_${dt}_t$t.bin NEDDF_PROBE_DTYPE=$dt python tools/pmc_probe.py 3 > $O/stamp_${dt}_t$t.log 2>&1
    echo "=== $dt, stamped tile ${t:-6}: colour kernel"
    python tools/stamp_timeline_col.py $O/col_${dt}_t$t.bin 3 $([ $dt = bf16 ] && echo 8 || echo 4) > $O/stamp_col_${dt}_t$t.txt 2>&1; head -3 $O/stamp_col_${dt}_t$t.txt; tail -9 $O/stamp_col_${dt}_t$t.txt
    echo "=== $dt, stamped tile ${t:-6}: distance kernel"
    python tools/stamp_timeline.py $O/ddf_${dt}_t$t.bin 7 > $O/stamp_ddf_${dt}_t$t.txt 2>&1; head -3 $O/stamp_ddf_${dt}_t$t.txt; tail -11 $O/stamp_ddf_${dt}_t$t.txt
  done
done
# launch times under the stamp build (stage timings of bench.py) against the shipped library in the same call
for lib in libneddf_hip_stamp.so libneddf_hip.so; do
  NEDDF_LIB_PATH=$ROOT/neddf_amd/csrc/$lib NEDDF_BENCH_PMC=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$lib.json 2>/dev/null
  python - $O/bench_$lib.json $lib <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], round(d["value"]), "rays/s  ddf launch %.2f ms  col launch %.2f ms" % (r["avg_launch_ms"], r["colour_kernel"]["avg_launch_ms"]))
PY
done
