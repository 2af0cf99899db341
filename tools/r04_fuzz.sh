#!/bin/bash
# round 4: the long fuzz runs beside the suite (final sources): every tagging layout, the filter / pair / bad-SUNK drop-ins (= the
# device-resident chain), the on-device record parser
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
timeout -k 10 420 python3 tools/fuzz_tag.py 300 ${FUZZ_SEED_TAG:-91000} > $o/fuzz_tag.log 2>&1; echo "fuzz_tag rc=$?"; tail -2 $o/fuzz_tag.log
timeout -k 10 330 python3 tools/fuzz_filters.py 300 ${FUZZ_SEED_FILT:-92000} > $o/fuzz_filters.log 2>&1; echo "fuzz_filters rc=$?"; tail -2 $o/fuzz_filters.log
timeout -k 10 200 python3 tools/fuzz_ingest.py 120 ${FUZZ_SEED_ING:-93000} > $o/fuzz_ingest.log 2>&1; echo "fuzz_ingest rc=$?"; tail -2 $o/fuzz_ingest.log
