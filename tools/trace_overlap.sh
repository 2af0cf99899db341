#!/bin/bash
# kernel trace of a short bench run (headline steps, the chain alone, then both halves pipelined): which kernels of the second half
# run beside the probe kernels, and what each costs the other (tools/trace_overlap.py reads the csv)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/${ROUND:-r05}; mkdir -p $o
rocprofv3 --kernel-trace -d $o/prof_trace -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-extra --no-engine > $o/trace_bench.json 2> $o/trace_bench.log || { tail -5 $o/trace_bench.log; exit 1; }
f=$(ls $o/prof_trace/*/x_kernel_trace.csv $o/prof_trace/x_kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_overlap.py $f > $o/trace_overlap.txt
rm -rf $o/prof_trace
tail -60 $o/trace_overlap.txt
