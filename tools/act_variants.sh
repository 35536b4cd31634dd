#!/bin/bash
# A/B libraries for the tanhExp evaluation mode of the fused kernels (tile_engine.h kActMode: 0 branch-exact, 1 closed form only,
# 2 closed form + small-argument polynomial): tools/bin/libneddf_hip_act{0,1,2}.so carry that mode under BOTH the fp32 and the
# split-fp16 policy.  Built here (hipcc cross-compiles), loaded on the GPU box with NEDDF_LIB_PATH=<file>.
set -e
cd "$(dirname "$0")/../neddf_amd/csrc"
mkdir -p ../../tools/bin
for m in 0 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-slp-vectorize \
      -DNEDDF_ACT_F32=$m -DNEDDF_ACT_SPLIT=$m -c field_kernels.hip -o ../../tools/bin/field_kernels.act$m.o &
done
wait
for m in 0 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libneddf_hip_act$m.so ../../tools/bin/field_kernels.act$m.o \
      neddf_capi.o render_kernels.o op_kernels.o train_kernels.o train_capi.o comm_capi.o -ldl
done
ls -la ../../tools/bin/libneddf_hip_act*.so
