#!/bin/bash
# usage: tools/pmc_r03.sh <tag> [bench args...]      SQ / TCC counters of the probe kernel (separate rocprofv3 --pmc passes, never
# combined with a trace) on a bench.py workload; GAVISUNK_LIB / GAVISUNK_MZ_M etc. are taken from the environment.
# -> gpurun_out/r03/<tag>_pmc.txt (per-launch means: a pass over config 3 is 8 launches)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/r03
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "FETCH_SIZE TCC_REQ_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  GAVISUNK_BENCH_ALLOW_KNOBS=1 rocprofv3 --pmc $grp -d gpurun_out/r03/pmc_${tag}_$i -o x --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config2 --no-extra --no-engine "$@" > gpurun_out/r03/pmc_${tag}_$i.log 2>&1 || { tail -5 gpurun_out/r03/pmc_${tag}_$i.log; exit 1; }
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/r03/pmc_${tag}_* | tee gpurun_out/r03/${tag}_pmc.txt
rm -rf gpurun_out/r03/pmc_${tag}_*
