# dev tool (GPU box): host-side knobs of `kmdiff-hip diff` on one fabricated run directory
python3 tools/cli_throughput.py --rows 2000000 --parts 12 --keep /tmp/kmrun_keep 2>&1 | grep -E "run dir"
for x in "" "--matrix-path"; do
  for t in 64; do
    echo "flags: $x -t $t"
    KMD_HOST_TIMING=1 kmdiff_amd/bin/kmdiff-hip diff -d /tmp/kmrun_keep/km -1 20 -2 20 -o /tmp/kmrun_keep/out$t -t $t $x 2>&1 | grep -E "steady|waited|Partitions processed|significant" | cut -c1-200
  done
done
rm -rf /tmp/kmrun_keep
