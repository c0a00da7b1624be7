#!/bin/bash
# Build tuning variants of kmd_filter.hip into build_sweep/<name>.so (dev tool).
# usage: tools/sweep.sh name "-DKMD_BATCH=4 -DKMD_RPL_U32=2" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p build_sweep
OBJ=kmdiff_amd/lib/obj
build_one() {
  name=$1; flags=$2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags \
     -Rpass-analysis=kernel-resource-usage -c kmdiff_amd/csrc/kmd_filter.hip -o build_sweep/$name.o 2> build_sweep/$name.log
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_sweep/$name.so build_sweep/$name.o $OBJ/kmd_api.o $OBJ/kmd_correct.o
  grep -A8 "k_filter_soaIjLi" build_sweep/$name.log | grep -E "VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: [^ ]* *//; s/\[-R.*//' | tr '\n' ' '
  echo " <- $name"
}
while [ $# -gt 1 ]; do
  build_one "$1" "$2" &
  shift 2
  if [ $(jobs -r | wc -l) -ge 4 ]; then wait -n; fi
done
wait
