run() { timeout 200 python3 "$@" 2>/dev/null < /dev/null | tail -1 | cut -c1-125; }
for lib in flat2 flat3; do for kb in 4 8 16; do
  echo "== $lib tile $kb KB"
  KMD_LIB=$PWD/build_sweep/$lib.so KMD_FLAT_TILE_KB=$kb run tools/kbench.py --iters 20 --tag r21 --layout rows --nc 21 --nk 21
  KMD_LIB=$PWD/build_sweep/$lib.so KMD_FLAT_TILE_KB=$kb run tools/kbench.py --iters 20 --tag u8 --layout rows --count-bytes 1
  KMD_LIB=$PWD/build_sweep/$lib.so KMD_FLAT_TILE_KB=$kb run tools/kbench.py --iters 20 --tag r60v61 --layout rows --nc 60 --nk 61 --rows 16000000
done; done
