# A/B inside ONE gpurun call (boxes differ by +-2.5 %): bench lines or kernel stats of a library variant against the tree's build.
# Build the variant by hand (e.g. hipcc -c the edited file to /tmp/x.o, link it with the tree's other objects into
# tools/bin/libneddf_hip_<X>.so -- git-ignored, travels with gpurun) and select it with NEDDF_LIB_PATH.
O=$PWD/gpurun_out/quick; mkdir -p $O; export TMPDIR=/tmp; ROOT=$PWD
VARIANT=${1:-X}; ARGS=${2:---dtype f16_split}
for rep in 1 2; do
for v in tree $VARIANT; do
if [ $v = tree ]; then unset NEDDF_LIB_PATH; else export NEDDF_LIB_PATH=$ROOT/tools/bin/libneddf_hip_$v.so; fi
python bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']), round(d['ms_per_step'],1), r.get('avg_launch_ms'), (r.get('colour_kernel') or {}).get('avg_launch_ms'), d.get('psnr_vs_oracle_db'))"
done; done
