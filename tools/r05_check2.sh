#!/bin/bash
mkdir -p gpurun_out
python -m pytest -x -q -m gpu tests/test_gpu_cli.py::test_cli_more_near_threshold_rows_than_one_launch_lists \
  tests/test_gpu_bench.py::test_bench_eight_ranks_folded_onto_this_gpu tests/test_gpu_bench.py::test_bench_a_rank_that_dies_ends_the_job \
  tests/test_gpu_bench.py::test_two_ranks_through_rccl_on_this_box tests/test_gpu_threshold.py tests/test_gpu_refine.py \
  -rs 2>&1 | tail -40 > gpurun_out/r05_check2.txt
cat gpurun_out/r05_check2.txt
tools/r05_sparse_ab.sh > gpurun_out/r05_sparse_ab.log 2>&1
cat gpurun_out/sparse_ab/summary.txt
python tests/soak.py --only popstrat --popstrat-stand true --tally --seconds 300 --seed 51 > gpurun_out/soak_ps_stand.txt 2>&1
tail -8 gpurun_out/soak_ps_stand.txt
