#!/bin/bash
# Times tools/bgzf_time.py (the GPU BGZF reader alone) for the product library and every libgavisunk_<variant>.so named:
#   tools/bgzf_variants.sh <out.log> <GB> <level> [variant ...]      (variants are built with python -m gavisunk_amd.build --variant)
out=$1; gb=$2; level=$3; shift 3
: > "$out"
for v in product "$@"; do
    if [ "$v" = product ]; then unset GAVISUNK_LIB; else export GAVISUNK_LIB=$PWD/gavisunk_amd/libgavisunk_$v.so; fi
    echo "== $v" >> "$out"
    GAVISUNK_TIMING=1 timeout -k 10 250 python tools/bgzf_time.py "$gb" "$level" 2>&1 | grep -v "rowwriter" | tail -n 4 >> "$out" || exit 1
done
