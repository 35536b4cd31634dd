#!/bin/bash
# Same-call A/B of library variants on the TRAINING step: tools/r06_ab_train.sh <out-tag> <dtype: f32 | f16_split> <variant> [<variant> ...]   ("main" = the shipped library)
ROOT=$PWD
TAG=$1; DTYPE=$2; shift 2
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset NEDDF_LIB_PATH; else export NEDDF_LIB_PATH=$ROOT/neddf_amd/csrc/libneddf_hip_v_$v.so; fi
  python bench.py --workload train --dtype $DTYPE --steps 10 --warmup 3 --no-cpu-baseline > $O/train_${v}_$rep.json 2> $O/train_${v}_$rep.err || tail -5 $O/train_${v}_$rep.err
done
done
unset NEDDF_LIB_PATH
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/train_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %9.1f rays/s  ms/step %8.3f  final_loss %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d.get("final_loss")))
    except Exception as e: print(f, "ERR", e)
PY
