#!/bin/bash
# tools/microbench/table_store_flavours.hip under rocprofv3: time, WRITE_SIZE and FETCH_SIZE per store flavour
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/mb_store
rm -rf $out; mkdir -p $out
B=$R/tools/microbench/build/mb_store
[ -x $B ] || hipcc --offload-arch=gfx950 -O2 -w -o $B $R/tools/microbench/table_store_flavours.hip
$B > $out/plain.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- $B > $out/w.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- $B > $out/f.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print(open("$out/plain.txt").read())
for k in sorted(agg):
    d = agg[k]; n = max(cnt[k].values())
    print("%-28s per launch: WRITE_SIZE %.2f GB  FETCH_SIZE %.2f GB" % (k, d["WRITE_SIZE"] / n * 1024 / 1e9, d["FETCH_SIZE"] / n * 1024 / 1e9))
PY
