#!/bin/bash
# Round-5 closing pass (on the GPU box, from the repository root): the whole -m gpu suite under the driver's conditions, smoke(),
# the colour kernel's phase stamps under the three operand policies (make stamp; tools/stamp_timeline_col.py), the default bench line.
ROOT=$PWD
O=$ROOT/gpurun_out/r5final
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
python __graft_entry__.py smoke-only > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
if [ -f neddf_amd/csrc/libneddf_hip_stamp.so ]; then
  for dt in fp32 f16_split bf16; do
    NEDDF_LIB_PATH=$ROOT/neddf_amd/csrc/libneddf_hip_stamp.so NEDDF_STAMP_FILE=$O/ddf_$dt.bin NEDDF_STAMP_FILE_COL=$O/col_$dt.bin NEDDF_PROBE_DTYPE=$dt \
      python tools/pmc_probe.py 2 > $O/stamp_$dt.log 2>&1
    echo "=== colour kernel, $dt"
    python tools/stamp_timeline_col.py $O/col_$dt.bin 3 $([ $dt = bf16 ] && echo 8 || echo 4) | tee $O/stamp_col_$dt.txt
    python tools/stamp_timeline.py $O/ddf_$dt.bin 7 > $O/stamp_ddf_$dt.txt 2>&1
    rm -f $O/ddf_$dt.bin $O/col_$dt.bin
  done
fi
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5final/bench_default.json")); r = d["roofline"]
print(round(d["value"]), d["unit"], "ms/step %.2f" % d["ms_per_step"], "frac %.3f" % r["frac"], "traffic", r.get("traffic"), "col ms", r["colour_kernel"]["avg_launch_ms"], "psnr", d.get("psnr_vs_oracle_db"))
PY
