#!/bin/bash
# dev tool (GPU box): records per tile vs time (is the number of tiles per workgroup an integer?)
cd $GRAFT_REPO_ROOT
for f in 0 33900 25385 25000 20300 20000 16900; do
  echo "== KMD_TILE_FILL=$f"
  KMD_TILE_FILL=$f KMD_DEBUG=1 timeout 100 python tools/kbench_pipeline.py --fused-only --iters 1 2>&1 | grep "level 0" | head -1 | sed 's/.*level 0: //'
  KMD_TILE_FILL=$f timeout 300 python tools/kbench_pipeline.py --fused-only --iters 6 2>&1 | grep fused | tail -1 | sed 's/.*kmd_merge_filter) //'
done
