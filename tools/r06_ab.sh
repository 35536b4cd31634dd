#!/bin/bash
# Same-call A/B of library variants (make variant NAME=.. DEFS=..): bench lines of one operand policy per variant, on ONE box.
#   tools/r06_ab.sh <out-tag> <dtype> <variant> [<variant> ...]      ("main" = the shipped libneddf_hip.so)
ROOT=$PWD
TAG=$1; DTYPE=$2; shift 2
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0 NEDDF_BENCH_PROBE=${NEDDF_BENCH_PROBE:-0}
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset NEDDF_LIB_PATH; else export NEDDF_LIB_PATH=$ROOT/neddf_amd/csrc/libneddf_hip_v_$v.so; fi
  python bench.py --dtype $DTYPE --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err || tail -5 $O/bench_${v}_$rep.err
done
done
unset NEDDF_LIB_PATH
python - "$O" <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        st=d.get("stage_ms_per_step",{})
        print("%-28s %9d rays/s  ms/step %8.2f  ddf %8.2f col %7.2f  frac %.4f  psnr %s" % (os.path.basename(f), d["value"], d["ms_per_step"], st.get("ddf",0), st.get("col",0), r.get("frac",0), d.get("psnr_vs_oracle_db")))
    except Exception as e: print(f, "ERR", e)
PY
