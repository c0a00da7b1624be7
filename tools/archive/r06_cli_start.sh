#!/bin/bash
# dev (GPU box): where the first partitions' time goes (KMD_HOST_TIMING=2: a line per partition)
repo=${GRAFT_REPO_ROOT:-$PWD}
cd "$repo"
python3 tools/cli_throughput.py --rows 2000000 --parts 8 --keep /tmp/kmd_run > /dev/null 2>&1
ls /tmp/kmd_run | head -3
for i in 1 2; do
  KMD_HOST_TIMING=2 kmdiff_amd/bin/kmdiff-hip diff -d /tmp/kmd_run/km -1 20 -2 20 -o /tmp/kmd_out_$i -t 16 2>&1 | grep -E "partition [0-9]+ d|worker ready|workers done|Done in|Partitions processed|last partition"
  echo
done
