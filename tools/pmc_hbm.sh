#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the compress kernels for the current environment (two PMC passes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_hbm
rm -rf $out; mkdir -p $out
cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 tools/compress_stage_times.py > $out/log_f.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 tools/compress_stage_times.py > $out/log_w.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:24]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in ("mts::k_match5", "mts::k_hash_sort"):
    d = agg[k]; n = max(cnt[k].values()) if cnt[k] else 1
    print(k, "per launch: FETCH %.1f GB  WRITE %.1f GB" % (d["FETCH_SIZE"] / n / 1e6 * 1.024, d["WRITE_SIZE"] / n / 1e6 * 1.024))
print(open("$out/log_w.txt").read().strip().split("\n")[-1][:300])
PY
