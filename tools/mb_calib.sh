#!/bin/bash
# tools/microbench/hbm_counter_calibration.hip under rocprofv3: FETCH_SIZE / WRITE_SIZE against known byte counts
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/mb_calib
rm -rf $out; mkdir -p $out $R/tools/microbench/build
B=$R/tools/microbench/build/mb_calib
[ -x $B ] || hipcc --offload-arch=gfx950 -O2 -w -o $B $R/tools/microbench/hbm_counter_calibration.hip
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- $B > $out/f.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- $B > $out/w.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"]) * 1024
B = 4 << 30; L = B // 128
known = {"k_read16_stream": ("FETCH_SIZE", B), "k_read4_stream": ("FETCH_SIZE", B), "k_read16_gather": ("FETCH_SIZE", L * 128),
         "k_write16_stream": ("WRITE_SIZE", B), "k_write4_stream": ("WRITE_SIZE", B), "k_write4_scatter": ("WRITE_SIZE", L * 128)}
print(open("$out/f.txt").read().strip().split("\n")[-1])
for k, (c, b) in known.items():
    v = agg[k][c]
    print("%-18s %s = %.3f GB for %.3f GB (lines; the other counter: %.3f GB)  counter / bytes = %.3f" % (k, c, v / 1e9, b / 1e9, agg[k]["WRITE_SIZE" if c == "FETCH_SIZE" else "FETCH_SIZE"] / 1e9, v / b))
PY
