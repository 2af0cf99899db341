# usage: bash tools/final_profile_r02.sh <tag>      round-2 evidence on the headline workload (config 3, full size):
#   gpurun_out/r02/<tag>_bench.json                 the bench line (default steps, with parity + CPU baseline + config-2 leg)
#   gpurun_out/r02/<tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command (short run)
#   gpurun_out/r02/<tag>_pmc.txt                    SQ / TCC counters of the probe kernel (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1
mkdir -p gpurun_out/r02
python3 bench.py > gpurun_out/r02/${tag}_bench.json 2> gpurun_out/r02/${tag}_bench.log || { tail -5 gpurun_out/r02/${tag}_bench.log; exit 1; }
tail -4 gpurun_out/r02/${tag}_bench.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/prof_$tag -o x --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config2 > gpurun_out/r02/${tag}_bench_under_rocprof.json 2> gpurun_out/r02/prof_$tag.log || { tail -5 gpurun_out/r02/prof_$tag.log; exit 1; }
f=$(ls gpurun_out/r02/prof_$tag/*/x_kernel_stats.csv gpurun_out/r02/prof_$tag/x_kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/r02/${tag}_kernel_stats.csv
rm -rf gpurun_out/r02/prof_$tag
python3 tools/kstats.py gpurun_out/r02/${tag}_kernel_stats.csv 2>/dev/null | head -8
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "FETCH_SIZE TCC_REQ_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $grp -d gpurun_out/r02/pmc_${tag}_$i -o x --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config2 > gpurun_out/r02/pmc_${tag}_$i.log 2>&1 || { tail -5 gpurun_out/r02/pmc_${tag}_$i.log; exit 1; }
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/r02/pmc_${tag}_* | tee gpurun_out/r02/${tag}_pmc.txt
rm -rf gpurun_out/r02/pmc_${tag}_*
