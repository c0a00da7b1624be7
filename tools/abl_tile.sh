cd $GRAFT_REPO_ROOT
run() { echo "== $* $ARGS"; env "$@" timeout 300 python tools/kbench_pipeline.py --fused-only --iters 4 $ARGS 2>&1 | grep -E "fused|Error|error" | tail -1; }
(timeout 400 python -m pytest tests/test_gpu_tilemerge.py -q -x 2>&1 | tail -5)
ARGS="" run KMD_TILE_XCD=1
ARGS="" run KMD_LIB=build_sweep/tile_r4.so
ARGS="" run KMD_LIB=build_sweep/tile_r12.so
ARGS="" run KMD_LIB=build_sweep/tile_r16.so
ARGS="" run KMD_TILE_SHAPE=1024x4096
ARGS="" run KMD_TILE_SHAPE=1024x2048
ARGS="" run KMD_TILE_LOAD_PCT=25
ARGS="" run KMD_TILE_XCD=0
ARGS="--sparse 0.3" run KMD_TILE_XCD=1
ARGS="--sparse 0.1" run KMD_TILE_XCD=1
ARGS="--nc 4 --nk 4" run KMD_TILE_XCD=1
ARGS="--nc 100 --nk 100 --rows 1000000" run KMD_TILE_XCD=1
ARGS="--keys clustered" run KMD_TILE_XCD=1
bash tools/prof_tile.sh
bash tools/pmc_tile.sh
