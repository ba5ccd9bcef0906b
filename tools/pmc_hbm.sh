#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per kernel (two PMC passes):  tools/pmc_hbm.sh [script] [kernel name filter]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_hbm
rm -rf $out; mkdir -p $out
cd $R
S=${1:-tools/compress_stage_times.py}
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 $S > $out/log_f.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 $S > $out/log_w.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -(agg[k]["FETCH_SIZE"] + agg[k]["WRITE_SIZE"]) / max(cnt[k].values())):
    if "${2:-mts}" not in k: continue
    d = agg[k]; n = max(cnt[k].values())
    print("%-42s launches %3d  per launch: FETCH %7.2f GB  WRITE %7.2f GB" % (k, n, d["FETCH_SIZE"] / n * 1024 / 1e9, d["WRITE_SIZE"] / n * 1024 / 1e9))
PY
