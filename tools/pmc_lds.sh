#!/bin/bash
# LDS / issue counters of every kernel of one compress pass (rocprofv3 PMC passes, no tracing domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_lds
rm -rf $out; mkdir -p $out
cd $R
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/a -- python3 ${1:-tools/compress_stage_times.py} > $out/log_a.txt 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $out/b -- python3 ${1:-tools/compress_stage_times.py} > $out/log_b.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0][:30]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0))[:6]:
    print(k, {n: "%.3e" % v for n, v in sorted(d.items())})
PY
