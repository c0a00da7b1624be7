for v in ord1 ord2; do echo "== parity $v"; KMD_LIB=$PWD/build_sweep/$v.so timeout 1200 python -m pytest tests/test_gpu_tilemerge.py tests/test_gpu_threshold.py -q -m gpu -x 2>&1 | tail -2; done
bash tools/r06_ab.sh -s "4 3 1 5 6 2" build_sweep/ord0.so build_sweep/ord1.so build_sweep/ord2.so
bash tools/pmc_ab.sh -a "--rows 39062500" -k k_tile_sums build_sweep/ord0.so build_sweep/ord1.so build_sweep/ord2.so 2>&1 | tail -40
