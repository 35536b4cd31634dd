#!/bin/bash
# phase stamps of the reverse-mode distance kernel (make stamp build) for one operand policy: tools/r06_stamp.sh <out-tag> <dtype> [lib suffix]
ROOT=$PWD
O=$ROOT/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
LIB=$ROOT/neddf_amd/csrc/libneddf_hip_${3:-stamp}.so
NEDDF_LIB_PATH=$LIB NEDDF_STAMP_FILE=$O/st_$2.bin NEDDF_PROBE_DTYPE=$2 python tools/pmc_probe.py 2 > $O/probe_$2.log 2>&1
python tools/stamp_timeline.py $O/st_$2.bin > $O/stamp_timeline_$2.txt 2>&1
python tools/stamp_tiles.py $O/st_$2.bin > $O/stamp_tiles_$2.txt 2>&1
tail -22 $O/stamp_timeline_$2.txt; head -3 $O/stamp_tiles_$2.txt
