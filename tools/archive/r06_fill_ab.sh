#!/bin/bash
# dev: K2's matrix fill, variants of kmd_merge.hip side by side on one lease (parity first, then the kernels under rocprofv3)
# usage: tools/r06_fill_ab.sh <lib.so> ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r06_fill; mkdir -p $out
for lib in "$@"; do
  n=$(basename $lib .so)
  echo "== $n"
  KMD_LIB=$PWD/$lib timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -m gpu -k "merge or matrix" 2>&1 | tail -1
  for args in "--keys random" "--keys random --limbs 2" "--keys random --layout rows" "--keys random --sparse 0.1 --rows 8000000"; do
    rm -rf $out/t_$n
    KMD_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats -d $out/t_$n -o t --output-format csv -- python3 tools/kbench_merge.py $args > $out/$n.log 2>&1
    f=$(find $out/t_$n -name "t_kernel_stats.csv" | head -1)
    echo "  $args: $(grep -E "ms|rows" $out/$n.log | tail -1)"
    python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Name"] for k in ("k_fill_matrix", "k_row_windows", "k_tile_sums", "Radix", "radix", "k_gather", "k_iota"))]
print("     " + "; ".join("%s %sx %.1fus" % (r["Name"].split("(")[0].split("::")[-1][:40], r["Calls"], float(r["AverageNs"]) / 1e3) for r in keep[:8]))
PY
  done
done
