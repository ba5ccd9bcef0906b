#!/bin/bash
# Everything profiles/ holds for one round, from a 1-GPU box:  tools/profile_round.sh r1
#   <tag>_bench_default.jsonl      the default bench line
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command (per-kernel totals / averages)
#   <tag>_bench_pmc_hbm_bytes.csv  FETCH_SIZE / WRITE_SIZE per kernel (two separate --pmc passes, no trace domains)
#   <tag>_traffic.json             HBM bytes per launch of the dominant kernel, read by bench.py
#   <tag>_levels_kernel_stats.csv  rocprofv3 --kernel-trace --stats of tools/level_sweep.py (levels 1, 2, 3, 6, 7, 8, 9 on 240 chunks of 15.36 MB)
#   <tag>_levels.json              its output line
tag=${1:-r4}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/profile_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py > $out/${tag}_bench_default.jsonl 2> $out/bench_stderr.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-extras > $out/stats_stdout.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 > $out/pmc_fetch_stdout.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 > $out/pmc_write_stdout.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/levels -- python3 tools/level_sweep.py > $out/${tag}_levels.json 2> $out/levels_stderr.txt
python3 - <<PY
import csv, glob, json, collections
out, tag = "$out", "$tag"
f = glob.glob(out + "/levels/**/*kernel_stats.csv", recursive=True)
if f:
    open(out + "/%s_levels_kernel_stats.csv" % tag, "w").write(open(f[0]).read())
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
if f:
    open(out + "/%s_bench_kernel_stats.csv" % tag, "w").write(open(f[0]).read())
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for name, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for fn in glob.glob(out + "/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].split("(")[0]
            agg[k][name] += float(r["Counter_Value"]); n[k][name] += 1
with open(out + "/%s_bench_pmc_hbm_bytes.csv" % tag, "w") as w:
    w.write("kernel,launches,FETCH_SIZE_KB_total,WRITE_SIZE_KB_total\n")
    for k in sorted(agg, key=lambda k: -(agg[k]["FETCH_SIZE"] + agg[k]["WRITE_SIZE"])):
        w.write("%s,%d,%.3f,%.3f\n" % (k.replace(",", ";"), max(n[k].values()), agg[k]["FETCH_SIZE"], agg[k]["WRITE_SIZE"]))
per = {}
for k in agg:
    L = max(n[k].values())
    short = k.split("::")[-1].split("<")[0]
    # calibrated on known byte counts (tools/mb_calib.sh, round 4): FETCH_SIZE is HALF the bytes of the 128-B lines fetched, for streams
    # and for 16-byte gathers alike; WRITE_SIZE is exact, in 32-byte sectors
    per[short] = {"fetch_size_kb": agg[k]["FETCH_SIZE"] / L, "write_size_kb": agg[k]["WRITE_SIZE"] / L,
                  "traffic_bytes_per_launch": (2 * agg[k]["FETCH_SIZE"] + agg[k]["WRITE_SIZE"]) / L * 1024, "launches": L}
json.dump({"kernels": {k: v["traffic_bytes_per_launch"] for k, v in per.items()}, "detail": per,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 1 (60 x 23.1 MB chunks), bytes per launch = counter sum / launches x 1024; "
                  "traffic = 2 x FETCH_SIZE + WRITE_SIZE: calibrated with tools/microbench/hbm_counter_calibration.hip (tools/mb_calib.sh, profiles/r5_hbm_counter_calibration.txt) -- every read request "
                  "of an L2 to the fabric is a 128-B line (TCC_EA0_RDREQ x 128 B = the lines touched, TCC_EA0_RDREQ_32B = 0 in every pattern: 16-B streams, 4-B streams, one 16-B gather per line, "
                  "the 112-B stride of the old parse windows) and FETCH_SIZE tallies it as 64 B (its expression's 128-B term, TCC_BUBBLE, stays 0 on gfx950): x 2 is exact, not an estimate; WRITE_SIZE = "
                  "1.000 x for coalesced stores, 32 B per isolated 4-B store (sectors).  The figure is what the L2s ask of the fabric: lines that another XCD fetched a moment ago come out of the 256 MB "
                  "Infinity Cache and never reach HBM (k_parse_emit_marks in round 4: 2 x 4.94 + 1.21 GB in 1.53 ms = 7.25 TB/s, above the 5.6 TB/s a pure-read kernel and the 6.3 TB/s the 112-B stride "
                  "pattern reach in the microbenchmark -- a third of its lines were such re-fetches by other XCDs; with its workgroups of one segment group on one XCD it asks for 2 x 3.65 GB and takes the same time)",
           "calibration": {"fetch_factor": 2.0, "write_factor": 1.0, "write_granule_bytes": 32, "read_granule_bytes": 128},
           "source": "profiles/%s_bench_pmc_hbm_bytes.csv" % tag}, open(out + "/%s_traffic.json" % tag, "w"), indent=1)
print(open(out + "/%s_bench_default.jsonl" % tag).read()[:600])
PY
ls -la $out | head -20
