#!/bin/bash
# Round 5: weight / LDS fragment prefetch depth of the bf16 product pipeline (tools/bin/libneddf_hip_pf<DB><DA>.so), same call.
O=gpurun_out/r05p
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], round(d["value"]), "ms/step %.1f"%d["ms_per_step"], "ddf %.2f ms frac %.3f"%(r["avg_launch_ms"], r["frac"]), "col %.2f"%r["colour_kernel"]["avg_launch_ms"], "psnr %.1f"%d.get("psnr_vs_oracle_db"))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
for lib in base pf52 pf53 pf73; do
  NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_$lib.so timeout 300 python bench.py --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/b_$lib_$rep.json 2>$O/err.txt; line $O/b_$lib_$rep.json "bf16 $lib"
done
done
