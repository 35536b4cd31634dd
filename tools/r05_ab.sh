#!/bin/bash
# Round 5: same-call A/B of two libraries (tools/bin/libneddf_hip_base.so against $1) on the three operand policies of C2.
O=gpurun_out/r05f_$(basename $1 .so)
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
BASE=$PWD/tools/bin/libneddf_hip_base.so
NEW=$PWD/$1
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], round(d["value"]), "ms/step %.1f"%d["ms_per_step"], "ddf %.2f ms frac %.3f"%(r["avg_launch_ms"], r["frac"]), "col %.2f"%r["colour_kernel"]["avg_launch_ms"], "psnr %.1f"%d.get("psnr_vs_oracle_db"))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
for dt in ${DTYPES:-f32 bf16 f16_split}; do
  st=4; [ $dt = f32 ] && st=3
  NEDDF_LIB_PATH=$BASE timeout 300 python bench.py --dtype $dt --steps $st --warmup 1 --no-cpu-baseline > $O/b_${dt}_base_$rep.json 2>$O/err.txt; line $O/b_${dt}_base_$rep.json "$dt base"
  NEDDF_LIB_PATH=$NEW timeout 300 python bench.py --dtype $dt --steps $st --warmup 1 --no-cpu-baseline > $O/b_${dt}_new_$rep.json 2>$O/err.txt; line $O/b_${dt}_new_$rep.json "$dt new "
done
done
