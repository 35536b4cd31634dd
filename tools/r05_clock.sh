#!/bin/bash
# Round 5: which clock does the part hold under the field kernels?  (a) phase stamps of both field kernels with the constant 100 MHz
# clock beside the cycle counter (make stamp; tools/stamp_timeline.py, tools/stamp_timeline_col.py), (b) hwmon / amd-smi samples of
# sclk and socket power while bench.py runs (tools/clock_probe.py).
ROOT=$PWD
O=$ROOT/gpurun_out/r5clock
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for dt in fp32 f16_split bf16; do
  NEDDF_LIB_PATH=$ROOT/neddf_amd/csrc/libneddf_hip_stamp.so NEDDF_STAMP_FILE=$O/ddf_$dt.bin NEDDF_STAMP_FILE_COL=$O/col_$dt.bin NEDDF_PROBE_DTYPE=$dt \
    python tools/pmc_probe.py 3 > $O/stamp_$dt.log 2>&1
  echo "=== colour kernel, $dt"
  python tools/stamp_timeline_col.py $O/col_$dt.bin 3 $([ $dt = bf16 ] && echo 8 || echo 4) | tee $O/stamp_col_$dt.txt
  echo "=== distance kernel, $dt"
  python tools/stamp_timeline.py $O/ddf_$dt.bin 7 > $O/stamp_ddf_$dt.txt 2>&1; head -3 $O/stamp_ddf_$dt.txt; tail -12 $O/stamp_ddf_$dt.txt
done
python tools/clock_probe.py 6 > $O/clock_probe.json 2> $O/clock_probe.err; echo "clock probe rc=$?"
cat $O/clock_probe.json | head -150
amd-smi static -g 0 --limit --json > $O/amd_smi_limits.json 2>&1; head -40 $O/amd_smi_limits.json
