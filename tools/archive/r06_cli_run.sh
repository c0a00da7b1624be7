#!/bin/bash
# dev (GPU box): the command's throughput with this round's last two host changes (kmd_pack_records; the staging arrays
# released in the background) -- the full table, then stage 1 with the release back in the foreground, alternating
repo=${GRAFT_REPO_ROOT:-$PWD}
cd "$repo" && mkdir -p gpurun_out/r06 && O=gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_cli.py tests/test_gpu_pack.py -m gpu -x -q 2>&1 | tail -2
timeout 1200 python3 tools/cli_throughput.py --rows 2000000 --parts 8 --cpu-baseline > $O/cli_throughput.txt 2>&1 < /dev/null
tail -40 $O/cli_throughput.txt
{
  for i in 1 2 3; do
    echo "# background release (shipped)"; timeout 600 python3 tools/cli_throughput.py --rows 2000000 --parts 8 --only "-t 16" 2>&1 | grep -E "workers done|last partition|kmdiff-hip diff|Done in"
    echo "# KMD_SYNC_RELEASE=1: the staging arrays un-pinned before stage 1 ends (as until now)"; KMD_SYNC_RELEASE=1 timeout 600 python3 tools/cli_throughput.py --rows 2000000 --parts 8 --only "-t 16" 2>&1 | grep -E "workers done|last partition|kmdiff-hip diff|Done in"
  done
} > $O/cli_throughput_release.txt 2>&1
cat $O/cli_throughput_release.txt
