#!/bin/bash
# per-kernel average durations of one bench run (rocprofv3 --kernel-trace --stats), printed as a table
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/kstats
rm -rf $out; mkdir -p $out
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-extras > $out/stdout.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if float(r['AverageNs']) * int(r['Calls']) > 2e5:
        print("%-30s calls %3s avg %8.3f ms %6.2f%%" % (r['Name'].split('(')[0].replace('mts::', '')[:30], r['Calls'], float(r['AverageNs']) / 1e6, float(r['Percentage'])))
PY
