O=gpurun_out/r04/stamp; mkdir -p $O
NEDDF_REV_GEO_BF16=4x2x4 NEDDF_LIB_PATH=neddf_amd/csrc/libneddf_hip_stamp.so NEDDF_STAMP_FILE=$O/bf16_128.bin NEDDF_PROBE_DTYPE=bf16 python tools/pmc_probe.py 1 > $O/bf16_128.log 2>&1
python tools/stamp_timeline.py $O/bf16_128.bin 7 | tee $O/bf16_128.txt | head -80
