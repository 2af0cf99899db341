#!/bin/bash
# does the huge-page advice of the table loader's row arrays help or hurt once the page cache is full?  (A/B on one box, in turns)
mkdir -p gpurun_out/r04
for i in 1 2 3; do
  for thp in on off; do
    if [ $thp = off ]; then export GAVISUNK_NO_THP=1; else unset GAVISUNK_NO_THP; fi
    echo "== round $i, huge pages $thp"
    timeout -k 10 200 python3 tools/cli_1e8_time.py 2>&1 | grep -E "sunkdb\] loc:|run [0-9]" | cut -c1-200
    grep -E "AnonHugePages|^Cached|MemFree" /proc/meminfo | tr '\n' ' '; echo
  done
done
