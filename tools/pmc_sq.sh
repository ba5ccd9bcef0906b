#!/bin/bash
# SQ issue/wait counters of every kernel of one compress pass (rocprofv3 PMC pass, no tracing domains)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_sq
rm -rf $out; mkdir -p $out
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $out -- python3 ${1:-tools/compress_stage_times.py} ${@:2} > $out/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:22]:
    wc = d.get("SQ_WAVE_CYCLES", 1) or 1
    print("%-40s n=%3d wavecyc %.3e wait_any %.2f wait_inst %.2f active %.2f act_valu %.2f | insts valu %.3e vmem_rd %.3e lds %.3e" % (
        k, cnt[k], wc, d["SQ_WAIT_ANY"] / wc, d["SQ_WAIT_INST_ANY"] / wc, d["SQ_ACTIVE_INST_ANY"] / wc, d["SQ_ACTIVE_INST_VALU"] / wc,
        d["SQ_INSTS_VALU"], d["SQ_INSTS_VMEM_RD"], d["SQ_INSTS_LDS"]))
PY
