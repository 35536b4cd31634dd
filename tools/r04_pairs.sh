O=gpurun_out/r04/pairs; mkdir -p $O
for dt in f16_split bf16; do
  NEDDF_LIB_PATH=neddf_amd/csrc/libneddf_hip_stamppairs.so NEDDF_STAMP_FILE=$O/$dt.bin NEDDF_PROBE_DTYPE=$dt python tools/pmc_probe.py 1 > $O/$dt.log 2>&1
  echo "=== $dt"; python tools/stamp_pairs.py $O/$dt.bin 7 | tee $O/$dt.txt
done
