#!/bin/bash
# tools/build_variant.sh NAME [extra compiler flags]: a second build of the library (e.g. -DMTS_M5_TOPDOWN=0) as
# gpurun_scratch/lib_NAME.so, for tools/ab_stage_times.py.  Objects go to /tmp.
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/mts_variant_$name; mkdir -p $B $R/gpurun_scratch
for f in transform deflate match inflate api; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $R/mtscomp_amd/csrc/$f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/gpurun_scratch/lib_$name.so $B/transform.o $B/deflate.o $B/match.o $B/inflate.o $B/api.o
ls -la $R/gpurun_scratch/lib_$name.so
