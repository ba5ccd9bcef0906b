#!/bin/bash
# tools/pc_sample.sh [stochastic|host_trap] [interval] [library]: PC samples of one compress pass (rocprofv3 beta feature), summed per
# kernel and instruction (with the source line when the library was built with -gline-tables-only) into gpurun_out/pcs/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
method=${1:-stochastic}; interval=${2:-65536}; lib=$3
unit=cycles; [ "$method" = host_trap ] && unit=time
out=$R/gpurun_out/pcs
rm -rf $out; mkdir -p $out
cd $R
[ -n "$lib" ] && export MTSCOMP_HIP_LIB=$R/$lib
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $interval \
    --kernel-trace --output-format csv -d $out -- python3 tools/compress_stage_times.py > $out/log.txt 2>&1
echo "rocprofv3 exit $?" >> $out/log.txt
tail -5 $out/log.txt
ls -la $out/*/ 2>/dev/null | head -20
python3 tools/pc_sample_summary.py $out > $out/summary.txt 2>&1
head -60 $out/summary.txt
