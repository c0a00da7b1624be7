#!/bin/bash
# On the GPU box (gpurun -- 'bash tools/refresh_profiles_r06.sh [part ...]'): the evidence profiles/r06_* is built from, in
# ONE lease.  Parts (default: all): tests bench trace traffic c3 sparse ab kbench cli popstrat (hunt: on request)
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd "$repo" && mkdir -p gpurun_out/r06 && O=gpurun_out/r06
parts=${*:-tests bench trace traffic c3 sparse ab kbench cli popstrat}
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }

if has tests; then
  timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1 < /dev/null
  tail -2 $O/pytest_gpu.txt
  timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.txt 2>&1 < /dev/null
  tail -1 $O/pytest_gpu.txt
fi
if has bench; then
  timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_r06.json 2> $O/bench_r06.err < /dev/null
fi
if has trace; then
  # the same command under the kernel trace (without the legs that are not the headline): the CSV's average duration of
  # k_filter_soa against the HIP-event average the profiled run itself reports
  rm -rf $O/prof_bench
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline \
    > $O/prof_bench.json 2> $O/prof_bench.err < /dev/null
fi
if has traffic; then
  rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 tools/traffic_probe.py > $O/pmc_fetch.log 2>&1 < /dev/null
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 tools/traffic_probe.py > $O/pmc_write.log 2>&1 < /dev/null
fi
if has c3; then
  # kmd_merge_filter on a whole configs[2] partition: kernel trace, FETCH_SIZE / WRITE_SIZE, SQ counters of the merge kernel
  bash tools/prof_c3.sh $O/c3 > $O/c3_summary.txt 2>&1
  bash tools/pmc_ab.sh -a "--device --rows 39062500" -k "k_tile_sums<1024" kmdiff_amd/lib/libkmdiff_hip.so > $O/pmc_tile_sq.txt 2>&1      # (the plan takes the 4096-slot / 1024-thread shape at this size since round 5)
  rm -rf $O/prof_batch
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_batch -o batch -- python3 tools/kbench_batch.py --iters 2 --device --rows 39062500 --parts 6 \
    > $O/prof_batch.log 2>&1 < /dev/null
fi
if has sparse; then
  # rows of few records (2.9 / 7.8 per row) and the MIXED partition: kernel trace + SQ counters of the merge kernel
  rm -rf $O/prof_sparse $O/prof_mixed
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sparse -o sparse -- python3 tools/kbench_pipeline.py --fused-only --sparse 0.1 --rows 40000000 --iters 3 \
    > $O/prof_sparse.log 2>&1 < /dev/null
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mixed -o mixed -- python3 tools/kbench_pipeline.py --device --rows 39062500 --profile 1 --iters 5 \
    > $O/prof_mixed.log 2>&1 < /dev/null
  { bash tools/pmc_ab.sh -a "--fused-only --sparse 0.1 --rows 40000000 --iters 2" -k "k_tile_sums<1024" kmdiff_amd/lib/libkmdiff_hip.so
    bash tools/pmc_ab.sh -a "--fused-only --sparse 0.3 --rows 13333333 --iters 2" -k "k_tile_sums<512" kmdiff_amd/lib/libkmdiff_hip.so
    bash tools/pmc_ab.sh -a "--device --rows 39062500 --profile 1" -k "k_tile_sums<1024" kmdiff_amd/lib/libkmdiff_hip.so; } > $O/pmc_tile_sparse.txt 2>&1
fi
if has ab; then
  # the merge kernel before / after round 6's walk, same lease (build_sweep/r6_base2.so = round 5's kmd_tilemerge.hip with
  # this round's other objects; tools/r06_ab.sh)
  rm -rf gpurun_out/r06_ab
  bash tools/r06_ab.sh -s "1 2 3 4 5 6" build_sweep/r6_base2.so kmdiff_amd/lib/libkmdiff_hip.so > $O/ab_k2t.txt 2>&1
fi
if has hunt; then
  # the concurrent single-call path in fresh processes (round 4's abort: tests/test_gpu_tilemerge.py::test_first_filter_launch_on_fresh_streams)
  rm -rf gpurun_out/hunt
  bash tools/hunt_abort.sh 6 stress6 stress > $O/hunt.txt 2>&1
fi
if has kbench; then
  {
    run() { timeout 400 python3 "$@" 2>/dev/null < /dev/null | grep -E "fused|popstrat|merge\+filter|batch|pipeline" | tail -2; }
    run tools/kbench_pipeline.py --device --rows 39062500
    run tools/kbench_pipeline.py --device --rows 39062500 --partition 200
    run tools/kbench_pipeline.py --device --rows 16000000 --nc 50 --nk 50 --limbs 2
    run tools/kbench_pipeline.py --device --rows 100000000 --nc 4 --nk 4
    run tools/kbench_pipeline.py --device --rows 8000000 --nc 100 --nk 100
    run tools/kbench_pipeline.py --device --rows 4000000
    run tools/kbench_pipeline.py --device --rows 39062500 --profile 1
    for k in random even clustered; do run tools/kbench_pipeline.py --fused-only --keys $k; done
    run tools/kbench_pipeline.py --fused-only --sparse 0.3 --rows 13333333 --iters 3
    run tools/kbench_pipeline.py --fused-only --sparse 0.1 --rows 40000000 --iters 3
    run tools/kbench_pipeline.py --fused-only --limbs 2
    run tools/kbench_batch.py --device --rows 39062500 --parts 6
    run tools/kbench_batch.py --device --rows 39062500 --parts 12
    run tools/kbench_batch.py --parts 12
    run tools/kbench_popstrat.py
    run tools/kbench_popstrat.py --thr 0.05
    # the matrix path's merge (kmd_merge_partition: the tile merge + the matrix fill) beside the fused call
    run tools/kbench_pipeline.py --iters 4
    run tools/kbench_pipeline.py --iters 4 --nc 50 --nk 50
    run tools/kbench_pipeline.py --iters 4 --sparse 0.1 --rows 20000000
  } > $O/kbench.txt 2>&1
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
    bash tools/kbench_k1r.sh
  } > $O/kbench_k1.txt 2>&1
fi
if has cli; then
  timeout 1200 python3 tools/cli_throughput.py --rows 2000000 --parts 8 --cpu-baseline > $O/cli_throughput.txt 2>&1 < /dev/null
  timeout 900 python3 tools/cli_throughput.py --rows 2000000 --parts 8 --ab 3 > $O/cli_throughput_ab.txt 2>&1 < /dev/null
  # the host packer of round 5 (build_sweep/r5_packer/libkmdiff_hip.so, LD_PRELOAD) against this round's, alternating, -t 16
  if [ -f build_sweep/r5_packer/libkmdiff_hip.so ]; then
    timeout 900 python3 tools/cli_throughput.py --rows 2000000 --parts 8 --ab 4 --ab-preload build_sweep/r5_packer/libkmdiff_hip.so > $O/cli_throughput_packer.txt 2>&1 < /dev/null
  fi
fi
if has popstrat; then
  bash tools/pmc_popstrat.sh --thr 0.05 > $O/pmc_popstrat.txt 2>&1
fi
ls $O
