#!/bin/bash
# dev tool (GPU box): K3 variants.  usage: bash tools/ab_popstrat.sh lib...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  for a in "--thr 0.05" "" "--nc 20 --nk 20 --thr 0.05"; do
    echo "== $v $a"; KMD_LIB=$v timeout 300 python tools/kbench_popstrat.py $a 2>&1 | grep popstrat | tail -1
  done
done
