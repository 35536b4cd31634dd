#!/bin/bash
# phase stamps of the colour kernel (make stamp build) for one operand policy: tools/r06_stamp_col.sh <out-tag> <dtype> [waves per workgroup] [lib suffix]
ROOT=$PWD
O=$ROOT/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LIB=$ROOT/neddf_amd/csrc/libneddf_hip_${4:-stamp}.so
NEDDF_LIB_PATH=$LIB NEDDF_STAMP_FILE_COL=$O/sc_$2.bin NEDDF_PROBE_DTYPE=$2 python tools/pmc_probe.py 2 > $O/probe_col_$2.log 2>&1
python tools/stamp_timeline_col.py $O/sc_$2.bin 3 ${3:-4} > $O/stamp_col_$2.txt 2>&1
python tools/stamp_tiles.py $O/sc_$2.bin > $O/stamp_tiles_col_$2.txt 2>&1
cat $O/stamp_col_$2.txt; head -3 $O/stamp_tiles_col_$2.txt
