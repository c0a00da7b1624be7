#!/bin/bash
# On the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): the evidence profiles/ is built from.
#   1. bench.py default run                          -> gpurun_out/bench_r01.json (+ soa / rows / BH variants)
#   2. rocprofv3 --kernel-trace --stats of the same  -> gpurun_out/prof_r01/bench_kernel_stats.csv
#   3. PMC passes (one counter set per pass, no trace domains) over tools/traffic_probe.py
#      -> gpurun_out/pmc_fetch, pmc_write, pmc_sq
# then, back in the container: python3 tools/summarize_profiles.py r01
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo" && mkdir -p gpurun_out
timeout 600 python3 bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err < /dev/null
timeout 300 python3 bench.py --no-cpu-baseline --layout soa > gpurun_out/bench_r01_soa.json 2>> gpurun_out/bench_r01.err < /dev/null
timeout 300 python3 bench.py --no-cpu-baseline --layout rows > gpurun_out/bench_r01_rows.json 2>> gpurun_out/bench_r01.err < /dev/null
timeout 300 python3 bench.py --no-cpu-baseline --correction benjamini > gpurun_out/bench_r01_bh.json 2>> gpurun_out/bench_r01.err < /dev/null
rm -rf gpurun_out/prof_r01 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o bench -- python3 bench.py --no-cpu-baseline \
  > gpurun_out/prof_bench.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 tools/traffic_probe.py \
  > gpurun_out/pmc_fetch.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 tools/traffic_probe.py \
  > gpurun_out/pmc_write.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d gpurun_out/pmc_sq -o pmc -- python3 tools/traffic_probe.py > gpurun_out/pmc_sq.log 2>&1 < /dev/null
ls gpurun_out/prof_r01 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
tail -c 600 gpurun_out/bench_r01.json
