#!/bin/bash
# gzip reader alone on the box: thread count x piece size (the box gives a process 16 CPUs of its 256 hardware threads)
mkdir -p gpurun_out/r04
for kb in 1024 2048 4096; do
  echo "== piece ${kb} KiB"
  GAVISUNK_GZ_PIECE_KB=$kb GZ_TIME_KEEP=1 GZ_TIME_THREADS=24,32,48,64 timeout -k 10 200 python3 tools/gz_time.py 6 2>&1 | grep threads
done
rm -f /tmp/gz_time.fq.gz
