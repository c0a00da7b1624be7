#!/bin/bash
# usage (on the GPU box): tools/prof_pmc.sh <outdir> "<pmc counters>" -- python3 tools/kbench.py ...
# Counter collection only (no tracing domains): gpurun refuses --pmc mixed with sys/hip traces.
out=$1; pmc=$2; shift 3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $pmc -d $GRAFT_REPO_ROOT/$out -o pmc --output-format csv -- "$@"
