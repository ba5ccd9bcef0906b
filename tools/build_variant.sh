#!/bin/bash
# tools/build_variant.sh NAME [extra compiler flags]: a second build of the library (e.g. -DMTS_M5_STATS=1) as
# gpurun_scratch/lib_NAME.so, for tools/ab_stage_times.py.  Same Makefile as the in-tree build (match.hip keeps its own flags); objects go to /tmp.
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_scratch
make -s -C $R/mtscomp_amd/csrc -j5 OUT=$R/gpurun_scratch/lib_$name.so BUILD=/tmp/mts_variant_$name \
  CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*"
ls -la $R/gpurun_scratch/lib_$name.so
