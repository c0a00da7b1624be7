#!/bin/bash
# dev (GPU box): SQ counters of the merge kernel for library variants on one shape + the phases of single tiles of a
# KMD_TILE_TIMING=2 build.   usage: tools/r06_pmc.sh "<kbench args>" "<kernel substring>" build_sweep/a.so ... [-- build_sweep/timing.so]
repo=${GRAFT_REPO_ROOT:-$PWD}
args=$1; kern=$2; shift 2
libs=(); timing=""
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then timing=$2; break; fi; libs+=("$1"); shift; done
bash tools/pmc_ab.sh -a "$args" -k "$kern" "${libs[@]}"
if [ -n "$timing" ]; then
  KMD_LIB=$repo/$timing timeout 300 python3 tools/kbench_pipeline.py --fused-only $args --iters 1 > gpurun_out/r06_timing.log 2>&1
  grep -h "tile phases" gpurun_out/r06_timing.log | tail -8
fi
