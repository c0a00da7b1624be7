run() { timeout 200 python3 "$@" 2>/dev/null < /dev/null | tail -1 | cut -c1-140; }
for lib in "$@"; do
  echo "== $lib"
  KMD_LIB=$PWD/build_sweep/$lib.so run tools/kbench.py --iters 20 --tag r20v20 --layout rows
  KMD_LIB=$PWD/build_sweep/$lib.so run tools/kbench.py --iters 20 --tag r24v24 --layout rows --nc 24 --nk 24
  KMD_LIB=$PWD/build_sweep/$lib.so run tools/kbench.py --iters 20 --tag r32v32 --layout rows --nc 32 --nk 32 --rows 20000000
  KMD_LIB=$PWD/build_sweep/$lib.so run tools/kbench.py --iters 20 --tag r34v34 --layout rows --nc 34 --nk 34 --rows 20000000
  KMD_LIB=$PWD/build_sweep/$lib.so run tools/kbench.py --iters 20 --tag r50v50 --layout rows --nc 50 --nk 50 --rows 16000000
  KMD_LIB=$PWD/build_sweep/$lib.so run tools/kbench.py --iters 20 --tag r100v100 --layout rows --nc 100 --nk 100 --rows 8000000
done
