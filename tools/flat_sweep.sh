# dev tool (GPU box): row-major any-pitch kernel variants.  usage: bash tools/flat_sweep.sh <variant> ...
for l in "$@"; do
  for a in "--nc 21 --nk 21" "--nc 3 --nk 3" "--nc 20 --nk 20 --count-bytes 1" "--nc 101 --nk 101" "--nc 21 --nk 21 --count-bytes 2"; do
    KMD_LIB=build_sweep/$l.so timeout 60 python3 tools/kbench.py --layout rows --rows 20000000 $a --tag "$l$(echo $a | tr -d ' -')" 2>&1 | tail -1 | cut -c1-30,85-125
  done
done
