#!/bin/bash
# usage: tools/final_profile_r04.sh <part>     round-4 evidence on the GPU box (results under gpurun_out/r04/, copied to profiles/r04/
# by tools/update_traffic.py): for each of the four GPU workloads -- config3 (the headline), diploid (BASELINE configs[3], one shard),
# config5 k = 21 and k = 25 (configs[4]) --
#   fin_<w>_bench.json          the workload's own bench line (HIP-event kernel time)
#   fin_<w>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of a short run of the same command
#   fin_<w>_pmc.txt             SQ / TCC counters of the probe kernel (separate --pmc passes of at most 8 counters, never with a trace)
# part = a (config3 + diploid) or b (config5 k21 + k25): two gpurun calls, each well inside the limit
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
run() {  # tag, bench args...
  tag=$1; shift
  python3 bench.py --steps 6 --warmup 1 --no-config2 --no-extra --no-engine --no-chain "$@" > $o/fin_${tag}_bench.json 2> $o/fin_${tag}_bench.log || { tail -5 $o/fin_${tag}_bench.log; return 1; }
  grep -h "Gbase" $o/fin_${tag}_bench.log | tail -1
  rocprofv3 --kernel-trace --stats -d $o/prof_$tag -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-extra --no-engine --no-chain "$@" > $o/fin_${tag}_bench_under_rocprof.json 2> $o/prof_$tag.log || { tail -5 $o/prof_$tag.log; return 1; }
  f=$(ls $o/prof_$tag/*/x_kernel_stats.csv $o/prof_$tag/x_kernel_stats.csv 2>/dev/null | head -1)
  cp $f $o/fin_${tag}_kernel_stats.csv
  rm -rf $o/prof_$tag
  python3 tools/kstats.py $o/fin_${tag}_kernel_stats.csv 2>/dev/null | head -4
  tools/pmc.sh r04 fin_$tag "$@" > /dev/null 2>&1
  cat $o/fin_${tag}_pmc.txt
}
if [ "$1" = a ]; then
  run config3 && run diploid --workload diploid
else
  run config5_k21 --workload config5 --k 21 && run config5_k25 --workload config5 --k 25
fi
