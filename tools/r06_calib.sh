#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the box (tools/traffic_calib.hip) + the list of counters rocprofv3 offers here
ROOT=$PWD
O=$ROOT/gpurun_out/${1:-calib}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -i -E "mall|dram|hbm|umc|EA0_RDREQ|EA0_WRREQ" $O/counters_available.txt | head -40 > $O/counters_memory_side.txt
$ROOT/tools/bin/traffic_calib > $O/calib_stdout.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $ROOT/tools/bin/traffic_calib > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $ROOT/tools/bin/traffic_calib > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $ROOT/tools/bin/traffic_calib > $O/stats.log 2>&1
cd $ROOT
python tools/traffic_calib.py $O $O/calib_stdout.txt $O/traffic_calib.json | tee $O/traffic_calib.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/calib_kernel_stats.csv 2>/dev/null
rm -rf $O/fetch $O/write $O/stats
cat $O/counters_memory_side.txt | head -20
