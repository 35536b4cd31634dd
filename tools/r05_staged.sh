#!/bin/bash
# Round 5: staged tile inputs + tail over every thread (ddf_rev_kernel, every policy) against the previous library, same call;
# rev2 at three workgroups per CU with y' as bf16 pairs / as eight bits.
O=gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
BASE=$PWD/tools/bin/libneddf_hip_base.so
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[2], round(d["value"]), "ms/step %.1f"%d["ms_per_step"], "ddf %.2f ms frac %.3f"%(r["avg_launch_ms"], r["frac"]), "col %.2f"%r["colour_kernel"]["avg_launch_ms"], "psnr %.1f"%d.get("psnr_vs_oracle_db"))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
for dt in f32 bf16 f16_split; do
  st=4; [ $dt = f32 ] && st=3
  NEDDF_LIB_PATH=$BASE timeout 300 python bench.py --dtype $dt --steps $st --warmup 1 --no-cpu-baseline > $O/b_${dt}_base_$rep.json 2>$O/err.txt; line $O/b_${dt}_base_$rep.json "$dt base  "
  timeout 300 python bench.py --dtype $dt --steps $st --warmup 1 --no-cpu-baseline > $O/b_${dt}_new_$rep.json 2>$O/err.txt; line $O/b_${dt}_new_$rep.json "$dt staged"
done
done
NEDDF_REV_GEO_BF16=2x3x4 NEDDF_REV2_Y16=1 timeout 300 python bench.py --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/b_rev2_y16.json 2>$O/err.txt; line $O/b_rev2_y16.json "bf16 rev2 2x3x4 y16"
NEDDF_REV_GEO_BF16=2x3x4 timeout 300 python bench.py --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/b_rev2_y8.json 2>$O/err.txt; line $O/b_rev2_y8.json "bf16 rev2 2x3x4 y8 "
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -x -q -m gpu -k "not subprocess and not random_arch and not exact_build" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
NEDDF_REV_GEO_BF16=2x3x4 NEDDF_REV2_Y16=1 timeout 600 python -m pytest tests/test_gpu_c5.py -x -q -m gpu > $O/pytest_y16.txt 2>&1; echo "pytest y16 rc=$?"; tail -2 $O/pytest_y16.txt
