O=gpurun_out/r04/stamp; mkdir -p $O
for dt in fp32 bf16; do
  for fu in 1 0; do
  NEDDF_FUSED=$fu NEDDF_LIB_PATH=neddf_amd/csrc/libneddf_hip_stamp.so NEDDF_STAMP_FILE=$O/${dt}_f$fu.bin NEDDF_PROBE_DTYPE=$dt python tools/pmc_probe.py 1 > $O/${dt}_f$fu.log 2>&1
  echo "=== $dt fused=$fu"; python tools/stamp_timeline.py $O/${dt}_f$fu.bin 7 > $O/${dt}_f$fu.txt; head -1 $O/${dt}_f$fu.txt; grep -E "colour|heads|tail|next-tile|encode " $O/${dt}_f$fu.txt; tail -12 $O/${dt}_f$fu.txt
  done
done
