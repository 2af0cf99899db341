#!/bin/bash
# round 4, closing: a long fuzz session on the final sources (seeds not used before)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
timeout -k 10 560 python3 tools/fuzz_tag.py 500 ${1:-110000} > $o/fuzzL_tag.log 2>&1; echo "fuzz_tag rc=$?"; tail -1 $o/fuzzL_tag.log
timeout -k 10 400 python3 tools/fuzz_filters.py 340 ${2:-120000} > $o/fuzzL_filters.log 2>&1; echo "fuzz_filters rc=$?"; tail -1 $o/fuzzL_filters.log
timeout -k 10 200 python3 tools/fuzz_ingest.py 150 ${3:-130000} > $o/fuzzL_ingest.log 2>&1; echo "fuzz_ingest rc=$?"; tail -1 $o/fuzzL_ingest.log
