#!/bin/bash
mkdir -p gpurun_out; rm -rf gpurun_out/sparse_ab
tools/r05_sparse_ab.sh > gpurun_out/r05_sparse_ab.log 2>&1
cat gpurun_out/sparse_ab/summary.txt
python tests/soak.py --only popstrat --popstrat-stand true --tally --seconds 120 --seed 52 > gpurun_out/soak_ps_stand2.txt 2>&1
grep "unexplained" gpurun_out/soak_ps_stand2.txt | head -20; tail -3 gpurun_out/soak_ps_stand2.txt
