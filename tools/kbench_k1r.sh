#!/bin/bash
# GPU box: K1r (row-major input) at the shapes whose pitch is not a multiple of 16 bytes, and its neighbours
run() { timeout 200 python3 "$@" 2>/dev/null < /dev/null | tail -1; }
kb() { tag=$1; shift; run tools/kbench.py --iters 20 --tag "$tag" "$@"; }
kb rows_21v21    --layout rows --nc 21 --nk 21
kb rows_3v3      --layout rows --nc 3 --nk 3 --rows 100000000
kb rows_u8_20v20 --layout rows --count-bytes 1
kb rows_u16_21v21 --layout rows --nc 21 --nk 21 --count-bytes 2
kb rows_10v11    --layout rows --nc 10 --nk 11
kb rows_60v61    --layout rows --nc 60 --nk 61 --rows 16000000
kb rows_20v20    --layout rows
kb rows_4v4      --layout rows --nc 4 --nk 4 --rows 100000000
