#!/bin/bash
# Round-3 probe 1 (GPU box, repository root): (a) does the lease expose compute partitions / more than one device,
# (b) what does the y' round trip of ddf_rev_kernel cost (timing-only probes of the -DNEDDF_ABLATE build), (c) what do the
# TCC counters report for each footprint.
ROOT=$PWD
O=$ROOT/gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
{
  echo "== nproc"; nproc; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"
  echo "== rocm-smi partitions"; timeout 60 rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -30
  echo "== amd-smi partition"; timeout 60 amd-smi partition 2>&1 | head -60
  echo "== sysfs"; for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition /sys/class/drm/card*/device/current_memory_partition; do echo "$f: $(cat $f 2>&1) [$(ls -l $f 2>&1 | cut -c1-10)]"; done
  echo "== /dev"; ls -l /dev/kfd /dev/dri 2>&1
  echo "== torch"; python -c "import torch; print('device_count', torch.cuda.device_count()); p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, p.total_memory, getattr(p,'L2_cache_size',None))"
  echo "== rocminfo agents"; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Name:.*gfx|L2:|L3:" | head -40
} > $O/partition_probe.txt 2>&1
export NEDDF_LIB_PATH=$ROOT/neddf_amd/csrc/libneddf_hip_ablate.so
for dt in fp32 bf16 f16_split; do
  for fl in 2 130 642 258; do
    NEDDF_PROBE_DTYPE=$dt NEDDF_SCHED=$fl timeout 300 python tools/ablate_probe.py 2>&1 | tail -1
  done
done > $O/yprime_probe.txt 2>&1
cd /tmp
for fl in 2 130 642 258; do
  for c in FETCH_SIZE WRITE_SIZE; do
    NEDDF_SCHED=$fl timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$fl/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_${fl}_$c.log 2>&1
  done
  python $ROOT/tools/pmc_summary.py $O/pmc_$fl > $O/pmc_summary_$fl.csv 2>&1
done
for fl in 2 642; do
  for c in FETCH_SIZE WRITE_SIZE; do
    NEDDF_PROBE_DTYPE=bf16 NEDDF_SCHED=$fl timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcbf_$fl/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/pmcbf_${fl}_$c.log 2>&1
  done
  python $ROOT/tools/pmc_summary.py $O/pmcbf_$fl > $O/pmcbf_summary_$fl.csv 2>&1
done
cd $ROOT
cat $O/partition_probe.txt $O/yprime_probe.txt
grep -h ddf_rev $O/pmc_summary_*.csv $O/pmcbf_summary_*.csv
