#!/bin/bash
# Round 5: how evenly do the workgroups of a persistent field launch progress?  (make stamp; tools/stamp_tiles.py)
ROOT=$PWD
O=$ROOT/gpurun_out/r5tiles
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for dt in fp32 bf16 f16_split; do
  NEDDF_LIB_PATH=$ROOT/neddf_amd/csrc/libneddf_hip_stamp.so NEDDF_STAMP_FILE=$O/ddf_$dt.bin NEDDF_STAMP_FILE_COL=$O/col_$dt.bin NEDDF_PROBE_DTYPE=$dt python tools/pmc_probe.py 3 > $O/stamp_$dt.log 2>&1
  echo "=== $dt distance kernel"; python tools/stamp_tiles.py $O/ddf_$dt.bin | tee $O/tiles_ddf_$dt.txt
  echo "=== $dt colour kernel"; python tools/stamp_tiles.py $O/col_$dt.bin | tee $O/tiles_col_$dt.txt
done
