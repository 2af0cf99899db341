#!/bin/bash
# usage: tools/final_profile_r03.sh <tag>      round-3 evidence (on the GPU box; results under gpurun_out/r03/, copied to profiles/r03/
# by tools/update_traffic.py <tag>):
#   <tag>_bench.json                the default bench line (config 3 headline + diploid shard + engine + config 2 + CLI legs)
#   <tag>_diploid_bench.json        BASELINE.json configs[3], one GPU's shard as the headline of its own line
#   <tag>_config5_k21/25_bench.json BASELINE.json configs[4]
#   <tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of a short config-3 run (8 probe launches per pass)
#   <tag>_pmc.txt                   SQ / TCC counters of the probe kernel (separate --pmc passes, never with a trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1
mkdir -p gpurun_out/r03
o=gpurun_out/r03
python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.log || { tail -5 $o/${tag}_bench.log; exit 1; }
grep -h "Gbase" $o/${tag}_bench.log
python3 bench.py --workload diploid --steps 6 --warmup 1 --no-config2 --no-extra --no-engine > $o/${tag}_diploid_bench.json 2> $o/${tag}_diploid_bench.log || { tail -5 $o/${tag}_diploid_bench.log; exit 1; }
for k in 21 25; do
  python3 bench.py --workload config5 --k $k --steps 6 --warmup 1 --no-config2 --no-extra --no-engine > $o/${tag}_config5_k${k}_bench.json 2> $o/${tag}_config5_k${k}_bench.log || { tail -5 $o/${tag}_config5_k${k}_bench.log; exit 1; }
done
grep -h "Gbase" $o/${tag}_diploid_bench.log $o/${tag}_config5_k21_bench.log $o/${tag}_config5_k25_bench.log
rocprofv3 --kernel-trace --stats -d $o/prof_$tag -o x --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config2 --no-extra --no-engine > $o/${tag}_bench_under_rocprof.json 2> $o/prof_$tag.log || { tail -5 $o/prof_$tag.log; exit 1; }
f=$(ls $o/prof_$tag/*/x_kernel_stats.csv $o/prof_$tag/x_kernel_stats.csv 2>/dev/null | head -1)
cp $f $o/${tag}_kernel_stats.csv
rm -rf $o/prof_$tag
python3 tools/kstats.py $o/${tag}_kernel_stats.csv 2>/dev/null | head -8
tools/pmc_r03.sh $tag > /dev/null 2>&1
cat $o/${tag}_pmc.txt
