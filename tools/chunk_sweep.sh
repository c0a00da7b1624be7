# dev tool (GPU box): host-side knobs of `kmdiff-hip diff` on one fabricated run directory
python3 tools/cli_throughput.py --rows 2000000 --parts 16 --keep /tmp/kmrun_keep 2>&1 | grep -E "run dir"
for d in 1 2 3; do
  for t in 256 64; do
    echo "ring depth $d, -t $t"
    KMD_RING_DEPTH=$d KMD_HOST_TIMING=1 kmdiff_amd/bin/kmdiff-hip diff -d /tmp/kmrun_keep/km -1 20 -2 20 -o /tmp/kmrun_keep/out -t $t 2>&1 | grep -E "steady|last part|Partitions processed" | cut -c1-200
  done
done
rm -rf /tmp/kmrun_keep
