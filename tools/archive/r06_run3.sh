#!/bin/bash
python -m pytest tests/test_gpu_threshold.py tests/test_gpu_bench.py tests/test_gpu_pack.py -x -q -m gpu 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_partition_size.py -x -q -m gpu 2>&1 | tail -6
for a in "" "--nc 50 --nk 50" "--sparse 0.1 --rows 20000000"; do python3 tools/kbench_pipeline.py --iters 4 $a 2>&1 | grep "merge+filter\|fused" ; done
KMD_LIB=$PWD/build_sweep/r6_base2.so python3 tools/kbench_pipeline.py --iters 4 2>&1 | grep "merge+filter"
python3 tools/kbench_merge.py 2>&1 | tail -8
