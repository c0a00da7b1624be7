#!/bin/bash
# Build a variant of one csrc file into build_sweep/<name>.so (dev tool); run it with KMD_LIB=build_sweep/<name>.so
# usage: tools/build_variant.sh <name> <file.hip> "<flags>"
set -e
cd "$(dirname "$0")/.."
mkdir -p build_sweep
name=$1; file=$2; flags=$3
OBJ=kmdiff_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags \
   -Rpass-analysis=kernel-resource-usage -c kmdiff_amd/csrc/$file.hip -o build_sweep/$name.o 2> build_sweep/$name.log
others=""
for f in kmd_api kmd_filter kmd_correct kmd_shard kmd_pack kmd_popstrat kmd_merge kmd_tilemerge kmd_pca kmd_pack_host; do [ $f != $file ] && others="$others $OBJ/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_sweep/$name.so build_sweep/$name.o $others
echo "built build_sweep/$name.so"
