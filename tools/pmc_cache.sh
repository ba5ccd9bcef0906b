#!/bin/bash
# L2 / vector-L1 counters of every kernel of one compress pass (two rocprofv3 PMC passes, no tracing domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_cache
rm -rf $out; mkdir -p $out
cd $R
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $out/a -- python3 ${1:-tools/compress_stage_times.py} > $out/log_a.txt 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/b -- python3 ${1:-tools/compress_stage_times.py} > $out/log_b.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0][:36]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:8]:
    print(k, {n: "%.3e" % v for n, v in d.items()})
PY
