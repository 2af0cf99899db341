#!/bin/bash
# usage: tools/pmc.sh <round dir, e.g. r04> <tag> [bench args...]      SQ / TCC counters of the probe kernel (separate rocprofv3 --pmc
# passes of at most 8 counters, never combined with a trace) on a bench.py workload; GAVISUNK_LIB etc. are taken from the environment.
# -> gpurun_out/<round>/<tag>_pmc.txt (per-launch means over the launches of one warm-up pass and one timed pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rnd=$1; tag=$2; shift; shift
o=gpurun_out/$rnd
mkdir -p $o
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "FETCH_SIZE TCC_REQ_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  GAVISUNK_BENCH_ALLOW_KNOBS=1 rocprofv3 --pmc $grp -d $o/pmc_${tag}_$i -o x --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config2 --no-extra --no-engine "$@" > $o/pmc_${tag}_$i.log 2>&1 || { tail -5 $o/pmc_${tag}_$i.log; exit 1; }
  i=$((i+1))
done
python3 tools/pmc_summary.py $o/pmc_${tag}_* | tee $o/${tag}_pmc.txt
rm -rf $o/pmc_${tag}_*
