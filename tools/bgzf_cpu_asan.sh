#!/bin/bash
# csrc/gs_bgzf.hip -- the host code that walks bgzip's headers AND the kernels that inflate the members -- under AddressSanitizer +
# UBSan on the CPU: the product's source compiled behind tools/hip_cpu (a stand-in for the HIP runtime; the GPU pool refuses
# sanitizer runs), fed files of every block type and damaged copies (flipped bits, cuts, lying BSIZE / ISIZE fields).
# usage: bgzf_cpu_asan.sh [dir, default /tmp/bgzfcpu] [damaged copies, default 120]
set -e
cd "$(dirname "$0")/.."
D=${1:-/tmp/bgzfcpu}
N=${2:-120}
mkdir -p "$D"
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -Wno-unknown-attributes -Wno-ignored-attributes -I tools/hip_cpu -I include -I gavisunk_amd/csrc -o "$D/bgzf_cpu_main" \
    tools/bgzf_cpu_main.cpp gavisunk_amd/csrc/gs_host.cpp gavisunk_amd/csrc/gs_inflate.cpp -lz -pthread -ldl
python3 tools/bgzf_cpu_files.py "$D" "$N"
cd "$D"
GAVISUNK_BGZF_CHUNK_KB=512 ./bgzf_cpu_main l1.gz l6.gz l9.gz stored.gz fixed.gz huff.gz rle.gz mem1.gz noeof.gz fa.gz small.gz tiny.gz empty.gz empties.gz bad*.gz > run.log 2>&1 || { grep -v '^ok' run.log | head -40; echo "bgzf_cpu_asan: FAILED (see $D/run.log)"; exit 1; }
echo "bgzf_cpu_asan: $(grep -c '^ok' run.log) reads clean ($(grep -c 'here rc -' run.log) of them refused as damaged)"
