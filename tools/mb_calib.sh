#!/bin/bash
# tools/microbench/hbm_counter_calibration.hip under rocprofv3: FETCH_SIZE / WRITE_SIZE against known byte counts
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/mb_calib
rm -rf $out; mkdir -p $out $R/tools/microbench/build
B=$R/tools/microbench/build/mb_calib
hipcc --offload-arch=gfx950 -O2 -w -o $B $R/tools/microbench/hbm_counter_calibration.hip
$B > $out/times.txt 2>&1                                      # (un-profiled: the kernels' own times)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- $B > $out/f.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- $B > $out/w.txt 2>&1
# the request counters behind FETCH_SIZE (the guide: FETCH_SIZE = TCC_EA0_RDREQ x 64 B): all read requests, and the 32-byte ones
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/r -- $B > $out/r.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCC_BUBBLE|TCC_EA0_RD_UNCACHED" | head -20 > $out/counters_available.txt
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"]) * 1024
B = 4 << 30; L = B // 128
known = {"k_read16_stream": ("FETCH_SIZE", B), "k_read4_stream": ("FETCH_SIZE", B), "k_read16_gather": ("FETCH_SIZE", L * 128),
         "k_write16_stream": ("WRITE_SIZE", B), "k_write4_stream": ("WRITE_SIZE", B), "k_write4_scatter": ("WRITE_SIZE", L * 128),
         "k_read16_stride112": ("FETCH_SIZE", (B // 4096 - 1) * 69 * 128), "k_read16_sum": ("FETCH_SIZE", B), "k_copy16": ("FETCH_SIZE", B // 2)}
print(open("$out/times.txt").read())
for k, (c, b) in known.items():
    v = agg[k][c]
    rd, rd32 = agg[k]["TCC_EA0_RDREQ_sum"] / 1024, agg[k]["TCC_EA0_RDREQ_32B_sum"] / 1024
    print("%-18s %s = %.3f GB for %.3f GB (lines; the other counter: %.3f GB)  counter / bytes = %.3f   read requests %.3e (of them 32-byte %.3e): bytes / request = %.1f" %
          (k, c, v / 1e9, b / 1e9, agg[k]["WRITE_SIZE" if c == "FETCH_SIZE" else "FETCH_SIZE"] / 1e9, v / b, rd, rd32, b / rd if rd else 0))
PY
