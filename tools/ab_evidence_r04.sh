#!/bin/bash
# GPU box: the same-box A/B lines behind DESIGN.md 4 (round 4) -> gpurun_out/r04/ab_k2t.txt, ab_k1r.txt
# (variants built here: tools/build_variant.sh k2t_<name> kmd_tilemerge "<-D...>", k1r_depth<d> kmd_filter "-DKMD_FLAT_DEPTH=<d>")
repo=${GRAFT_REPO_ROOT:-$PWD}
cd "$repo" && mkdir -p gpurun_out/r04
{
  echo "# kmd_merge_filter on one whole configs[2] partition (1 014 558 591 records), library variants, same box, two rounds (tools/ab_tile3.sh)"
  echo "# nt = KMD_TILE_HINT 1 (round 3's loads), plain = 0 (default now), sc1 = 2, align = nt + KMD_TILE_ALIGN, jobptr = nt + KMD_TILE_JOB_PTR"
  L="build_sweep/k2t_nt.so build_sweep/k2t_plain.so build_sweep/k2t_sc1.so build_sweep/k2t_align.so build_sweep/k2t_jobptr.so"
  bash tools/ab_tile3.sh -a "--device --rows 39062500" $L $L 2>&1 | grep -E "^k2t_" | cut -c1-170
  echo "# HBM traffic of the merge kernel per variant (tools/pmc_fetch_ab.sh; algorithmic 1.2175e10 B)"
  bash tools/pmc_fetch_ab.sh -a "--device --rows 39062500" $L 2>&1 | grep -E "^k2t_"
} > gpurun_out/r04/ab_k2t.txt 2>&1
{
  echo "# K1r, any pitch (k_filter_rows_flat): spans in flight per wave (KMD_FLAT_DEPTH 1 / 2 / 3), same box (tools/kbench.py)"
  run() { timeout 200 python3 "$@" 2>/dev/null < /dev/null | tail -1 | cut -c1-150; }
  for d in 1 2 3; do
    for cfg in "rows_21v21 --nc 21 --nk 21" "rows_3v3 --nc 3 --nk 3 --rows 100000000" "rows_u8_20v20 --count-bytes 1" "rows_60v61 --nc 60 --nk 61 --rows 16000000"; do
      set -- $cfg; tag=$1; shift
      echo -n "depth $d  "; KMD_LIB=$repo/build_sweep/k1r_depth$d.so run tools/kbench.py --iters 20 --layout rows --tag $tag "$@"
    done
  done
} > gpurun_out/r04/ab_k1r.txt 2>&1
cat gpurun_out/r04/ab_k2t.txt gpurun_out/r04/ab_k1r.txt
