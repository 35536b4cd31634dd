#!/bin/bash
# Which AddressSanitizer options let the sanitizer build of the host side run next to the HIP runtime on this box?
# (AMD's compiler-rt intercepts hsa_amd_memory_pool_allocate.)  Prints one line per variant.
cd "$(dirname "$0")/.."
RT=$(make -s -C neddf_amd/csrc print-asan-rt)
LIB=$PWD/neddf_amd/csrc/libneddf_hip_asan.so
for opts in "detect_leaks=0" "detect_leaks=0:protect_shadow_gap=0" "detect_leaks=0:allocator_may_return_null=1" \
            "detect_leaks=0:max_allocation_size_mb=65536" "detect_leaks=0:quarantine_size_mb=16:malloc_context_size=2"; do
  for xn in "" "1"; do
    out=$(HSA_XNACK=$xn LD_PRELOAD=$RT ASAN_OPTIONS=$opts NEDDF_LIB_PATH=$LIB timeout 300 python __graft_entry__.py smoke-only 2>&1 | tail -3 | tr '\n' ' ')
    echo "ASAN_OPTIONS=$opts HSA_XNACK=$xn -> ${out:0:300}"
  done
done
