#!/bin/bash
# profiles/<tag>_counters.txt from a 1-GPU box: SQ, cache and HBM-byte counters of the compress kernels (separate PMC passes)
tag=${1:-r2}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
echo "# rocprofv3 PMC passes over tools/compress_stage_times.py (two compress passes of 60 x 23.1 MB chunks, 385 ch, level 6)"
echo "## SQ counters per launch (tools/pmc_sq2.sh)"; bash tools/pmc_sq2.sh 8 2>&1 | grep -E "^mts::"
echo "## L2 / L1 counters, totals of the run (tools/pmc_cache.sh)"; bash tools/pmc_cache.sh 2>&1 | grep -E "^mts::"
echo "## HBM bytes per launch (tools/pmc_hbm.sh)"; bash tools/pmc_hbm.sh 2>&1 | grep -E "^mts::"
} > gpurun_out/${tag}_counters_raw.txt
cat gpurun_out/${tag}_counters_raw.txt | cut -c1-200
