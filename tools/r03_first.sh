#!/bin/bash
# round 3, first GPU call: the test suite on the new tree, the K1 traffic probe (FETCH_SIZE / WRITE_SIZE passes of their
# own) and the K1 shape lines of this round's kbench file.
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo" && mkdir -p gpurun_out/r03 && O=gpurun_out/r03
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1 < /dev/null
tail -3 $O/pytest_gpu.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 tools/traffic_probe.py > $O/pmc_fetch.log 2>&1 < /dev/null
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 tools/traffic_probe.py > $O/pmc_write.log 2>&1 < /dev/null
{
  run() { timeout 200 python3 "$@" 2>/dev/null < /dev/null | tail -1; }
  kb() { tag=$1; shift; run tools/kbench.py --iters 20 --tag "$tag" "$@"; }
  kb tiled_20v20   --layout tiled
  kb tiled_4v4     --layout tiled --nc 4 --nk 4 --rows 100000000
  kb tiled_50v50   --layout tiled --nc 50 --nk 50 --rows 16000000
  kb tiled_100v100 --layout tiled --nc 100 --nk 100 --rows 8000000
  kb tiled_u16     --layout tiled --count-bytes 2
  kb tiled_u8      --layout tiled --count-bytes 1
  kb soa_20v20     --layout soa
  kb rows_20v20    --layout rows
  kb rows_4v4      --layout rows --nc 4 --nk 4 --rows 100000000
  kb rows_50v50    --layout rows --nc 50 --nk 50 --rows 16000000
  kb rows_100v100  --layout rows --nc 100 --nk 100 --rows 8000000
} > $O/kbench_k1.txt 2>&1
cat $O/kbench_k1.txt
