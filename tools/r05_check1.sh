#!/bin/bash
mkdir -p gpurun_out
python -m pytest -x -q -m gpu tests/test_gpu_popstrat.py::test_reference_linear_vectors_through_the_device_routines tests/test_gpu_popstrat.py::test_more_features_than_samples \
  "tests/test_gpu_parity.py::test_sharded_correction_a_rank_that_fails_does_not_hang_the_others" \
  tests/test_gpu_tilemerge.py::test_first_filter_launch_on_fresh_streams tests/test_gpu_tilemerge.py::test_merge_filter_partitions_in_flight \
  tests/test_gpu_cli.py::test_cli_partitions_sharded_over_gpus tests/test_gpu_cli.py::test_cli_a_rank_that_fails_ends_the_run_with_its_error \
  tests/test_gpu_cli.py::test_cli_more_near_threshold_rows_than_one_launch_lists \
  tests/test_gpu_bench.py::test_bench_eight_ranks_folded_onto_this_gpu tests/test_gpu_bench.py::test_bench_a_rank_that_dies_ends_the_job \
  tests/test_gpu_bench.py::test_two_ranks_through_rccl_on_this_box tests/test_gpu_threshold.py tests/test_gpu_refine.py \
  -rs 2>&1 | tail -40 > gpurun_out/r05_check1.txt
cat gpurun_out/r05_check1.txt
python tests/soak.py --only popstrat --popstrat-stand true --tally --seconds 300 --seed 51 > gpurun_out/soak_ps_stand.txt 2>&1
tail -5 gpurun_out/soak_ps_stand.txt
