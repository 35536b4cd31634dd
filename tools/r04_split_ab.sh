#!/bin/bash
# split-fp16 LDS row stride, same call / same box: tools/bin/libneddf_hip_stride8.so is the library built with the old row stride
# (2 (W + 8) halves = 8 dwords mod 64), the in-tree library has 12 mod 64.  Alternating runs.
for i in 1 2; do
  for lib in tools/bin/libneddf_hip_stride8.so neddf_amd/csrc/libneddf_hip.so; do
    NEDDF_LIB_PATH=$PWD/$lib python bench.py --dtype f16_split --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', round(d['value']), r['avg_launch_ms'], r['colour_kernel']['avg_launch_ms'])"
  done
done
