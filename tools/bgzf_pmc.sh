#!/bin/bash
# usage: tools/bgzf_pmc.sh <round dir> <tag> [GB] [level]      SQ / TCC counters of k_bgzf_inflate (separate rocprofv3 --pmc passes of at
# most 8 counters, never combined with a trace) on tools/bgzf_time.py; GAVISUNK_LIB is taken from the environment.
# -> gpurun_out/<round>/<tag>_pmc.txt (per-launch means)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rnd=$1; tag=$2; gb=${3:-10}; lv=${4:-1}
o=gpurun_out/$rnd
mkdir -p $o
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_IFETCH_LEVEL" "FETCH_SIZE TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $grp -d $o/pmc_${tag}_$i -o x --output-format csv -- python3 tools/bgzf_time.py $gb $lv > $o/pmc_${tag}_$i.log 2>&1 || { tail -5 $o/pmc_${tag}_$i.log; echo "(group $i failed: $grp)"; }
  i=$((i+1))
done
PMC_KERNELS=k_bgzf_inflate python3 tools/pmc_summary.py $o/pmc_${tag}_* | tee $o/${tag}_pmc.txt
rm -rf $o/pmc_${tag}_*
