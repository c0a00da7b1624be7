# dev tool (GPU box): K2 bucket capacity by sample count
for S2 in 4 8 10 16 20 32; do
  for c in 256 512 1024; do
    echo -n "S=$((2*S2)) cap=$c  "
    KMD_MERGE_CAP=$c KMD_MERGE_PATH=fast-only timeout 60 python3 tools/kbench_merge.py --iters 5 --keys random --nc $S2 --nk $S2 --rows $((80000000 / S2 / 2)) 2>&1 | tail -1 | cut -c60-110
  done
done
for S2 in 50 64; do
  for c in 512 1024; do
    echo -n "S=$((2*S2)) cap=$c  "
    KMD_MERGE_CAP=$c KMD_MERGE_PATH=fast-only timeout 60 python3 tools/kbench_merge.py --iters 5 --keys random --nc $S2 --nk $S2 --rows $((80000000 / S2 / 2)) 2>&1 | tail -1 | cut -c60-110
  done
done
