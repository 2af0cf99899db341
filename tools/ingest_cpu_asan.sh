#!/bin/bash
# csrc/gs_ingest.hip -- the device record parser: staging code AND kernels -- under AddressSanitizer + UBSan on the CPU: the product's
# source compiled behind tools/hip_cpu (a stand-in for the HIP runtime and the two rocprim calls; the GPU pool refuses sanitizer
# runs), over regular texts (held against the host record reader), irregular ones (to be refused) and randomly damaged copies.
# usage: ingest_cpu_asan.sh [dir, default /tmp/ingcpu] [mutated copies, default 60]
set -e
cd "$(dirname "$0")/.."
D=${1:-/tmp/ingcpu}
N=${2:-60}
mkdir -p "$D"
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -Wno-unknown-attributes -Wno-ignored-attributes -Wno-return-type-c-linkage -I tools/hip_cpu -I include -I gavisunk_amd/csrc \
    -o "$D/ingest_cpu_main" tools/ingest_cpu_main.cpp gavisunk_amd/csrc/gs_host.cpp gavisunk_amd/csrc/gs_inflate.cpp -lz -pthread -ldl
rm -f "$D"/*.txt
python3 tools/ingest_cpu_files.py "$D" "$N"
cd "$D"
# (the staging code keeps its pinned buffers and cached device blocks for the life of the process, by design: no leak report)
ASAN_OPTIONS=detect_leaks=0 ./ingest_cpu_main *.txt > run.log 2>&1 || { grep -v '^ok' run.log | cut -c1-400 | head -40; echo "ingest_cpu_asan: FAILED (see $D/run.log)"; exit 1; }
echo "ingest_cpu_asan: $(tail -1 run.log)"
