#!/bin/bash
# Idle time between the kernels of one compress + decompress pass (rocprofv3 --kernel-trace over tools/roundtrip_stage_times.py):
# per kernel its duration and the gap to the kernel before it on the device.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/gaps
rm -rf $out; mkdir -p $out
cd $R
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/roundtrip_stage_times.py 2 > $out/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last compress pass and the last decompress pass: from the last k_delta_rows on
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mts::", "")[:28] for r in rows]
last = max(i for i, n in enumerate(names) if n.startswith("k_delta_rows"))
prev_end = None; tot_gap = 0; tot = 0
for r, n in list(zip(rows, names))[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    if prev_end: tot_gap += max(gap, 0)
    tot += (e - s) / 1e3
    print("%-28s dur %9.1f us   gap before %8.1f us" % (n, (e - s) / 1e3, gap))
    prev_end = max(prev_end or 0, e)
print("kernels %.2f ms, gaps %.2f ms" % (tot / 1e3, tot_gap / 1e3))
PY
