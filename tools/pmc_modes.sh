cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for m in 2 6 4 3 0; do
  export GAVISUNK_DEBUG_MODE=$m
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d gpurun_out/pmcm_$m -o x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcm_$m.log 2>&1 || exit 1
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/time_m$m.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py gpurun_out/pmcm_2 gpurun_out/pmcm_6 gpurun_out/pmcm_4 gpurun_out/pmcm_3 gpurun_out/pmcm_0
grep -h kernel_ms gpurun_out/time_m*.log | cut -c1-400
