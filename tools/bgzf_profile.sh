#!/bin/bash
# rocprofv3 --kernel-trace --stats of the GPU BGZF reader alone (tools/bgzf_time.py: 10 GB of text, level 1 and level 6) and of the
# record parser on one 1.2 GB block (tools/ingest_time_big.py): gpurun_out/$ROUND/bgzf_level{1,6}_kernel_stats.csv,
# ingest_kernel_stats.csv -> profiles/$ROUND/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/${ROUND:-r05}; mkdir -p $o
for lv in 1 6; do
    rocprofv3 --kernel-trace --stats -d $o/prof_bgzf -o x --output-format csv -- python3 tools/bgzf_time.py 10 $lv > $o/bgzf_level${lv}_under_rocprof.log 2>&1 || { tail -5 $o/bgzf_level${lv}_under_rocprof.log; exit 1; }
    f=$(ls $o/prof_bgzf/*/x_kernel_stats.csv $o/prof_bgzf/x_kernel_stats.csv 2>/dev/null | head -1)
    cp $f $o/bgzf_level${lv}_kernel_stats.csv
    rm -rf $o/prof_bgzf
    GAVISUNK_TIMING=1 python3 tools/bgzf_time.py 10 $lv > $o/bgzf_level${lv}_time.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats -d $o/prof_ingest -o x --output-format csv -- python3 tools/ingest_time_big.py > $o/ingest_under_rocprof.log 2>&1 || { tail -5 $o/ingest_under_rocprof.log; exit 1; }
f=$(ls $o/prof_ingest/*/x_kernel_stats.csv $o/prof_ingest/x_kernel_stats.csv 2>/dev/null | head -1)
cp $f $o/ingest_kernel_stats.csv
rm -rf $o/prof_ingest
head -6 $o/bgzf_level1_kernel_stats.csv | cut -c1-200
head -8 $o/ingest_kernel_stats.csv | cut -c1-200
