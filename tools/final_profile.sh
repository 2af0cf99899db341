#!/bin/bash
# usage: tools/final_profile.sh <round, e.g. r05> <part>     a round's evidence on the GPU box, one gpurun call per part; results under
# gpurun_out/<round>/, copied to profiles/<round>/ (and profiles/<round>/traffic.json + README.md rewritten) by tools/update_traffic.py
#   part a      config3 (the headline) + diploid (BASELINE configs[3], one shard)
#   part b      config5 k = 21 and k = 25 (configs[4])
#   part c      config3 with --group-len 200 (location groups of ~157 SUNKs)
#   part chain  the second half: rocprofv3 --kernel-trace --stats of a run with the chain legs, PMC of its three largest kernels,
#               and the kernel trace of the both-halves loop (tools/trace_overlap.py)
# For each tagger workload:   fin_<w>_bench.json (the workload's own bench line, HIP-event kernel time),
#   fin_<w>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of a short run of the same command), fin_<w>_pmc.txt (SQ / TCC counters
#   of the probe kernel: separate --pmc passes of at most 8 counters, never with a trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rnd=$1; part=$2
o=gpurun_out/$rnd; mkdir -p $o
run() {  # tag, bench args...
  tag=$1; shift
  python3 bench.py --steps 6 --warmup 1 --no-config2 --no-extra --no-engine --no-chain "$@" > $o/fin_${tag}_bench.json 2> $o/fin_${tag}_bench.log || { tail -5 $o/fin_${tag}_bench.log; return 1; }
  grep -h "Gbase" $o/fin_${tag}_bench.log | tail -1
  rocprofv3 --kernel-trace --stats -d $o/prof_$tag -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config2 --no-extra --no-engine --no-chain "$@" > $o/fin_${tag}_bench_under_rocprof.json 2> $o/prof_$tag.log || { tail -5 $o/prof_$tag.log; return 1; }
  f=$(ls $o/prof_$tag/*/x_kernel_stats.csv $o/prof_$tag/x_kernel_stats.csv 2>/dev/null | head -1)
  cp $f $o/fin_${tag}_kernel_stats.csv
  rm -rf $o/prof_$tag
  python3 tools/kstats.py $o/fin_${tag}_kernel_stats.csv 2>/dev/null | head -4
  tools/pmc.sh $rnd fin_$tag --no-chain "$@" > /dev/null 2>&1
  cat $o/fin_${tag}_pmc.txt
}
case "$part" in
  a) run config3 && run diploid --workload diploid ;;
  b) run config5_k21 --workload config5 --k 21 && run config5_k25 --workload config5 --k 25 ;;
  c) run config3_group_len_200 --group-len 200 ;;
  chain)
    ROUND=$rnd bash tools/chain_profile.sh > $o/chain_profile.log 2>&1; tail -30 $o/chain_profile.log
    PMC_KERNELS=k_diag_lds,k_rows_from_hits,k_pair_graph,k_row_write tools/pmc.sh $rnd chain > /dev/null 2>&1; cat $o/chain_pmc.txt
    ROUND=$rnd bash tools/trace_overlap.sh > /dev/null 2>&1; head -30 $o/trace_overlap.txt ;;
  *) echo "part = a | b | c | chain"; exit 2 ;;
esac
