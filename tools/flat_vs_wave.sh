# dev tool (GPU box): aligned row-major shapes, the wave/wide kernels against the any-pitch (flat) kernel
for a in "--nc 20 --nk 20 --count-bytes 2" "--nc 24 --nk 24 --count-bytes 1" "--nc 32 --nk 32 --count-bytes 1" "--nc 100 --nk 100 --count-bytes 2" "--nc 100 --nk 100 --count-bytes 1" "--nc 8 --nk 8 --count-bytes 1"; do
  for f in 0 1; do
    if [ $f = 1 ]; then export KMD_ROWS_FLAT_ALL=1; else unset KMD_ROWS_FLAT_ALL; fi
    timeout 60 python3 tools/kbench.py --layout rows --rows 20000000 $a --tag "flat$f$(echo $a | tr -d ' -')" 2>&1 | tail -1 | cut -c1-36,85-125
  done
done
