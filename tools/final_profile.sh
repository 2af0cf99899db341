# round-end evidence: bench line, the same under rocprofv3 --kernel-trace --stats, PMC passes, counter calibration
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1
python3 bench.py --steps 30 --warmup 3 2>gpurun_out/${T}_bench.err | tail -1 > gpurun_out/${T}_bench.json || exit 1
rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_stats -o x --output-format csv -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_under_rocprof.log 2>&1 || exit 1
cp gpurun_out/${T}_stats/x_kernel_stats.csv gpurun_out/${T}_kernel_stats.csv
tail -1 gpurun_out/${T}_bench_under_rocprof.log | grep '^{' > gpurun_out/${T}_bench_under_rocprof.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  rocprofv3 --pmc $grp -d gpurun_out/${T}_pmc_$i -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_pmc_$i.log 2>&1 || exit 1
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/${T}_pmc_* > gpurun_out/${T}_pmc_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d gpurun_out/${T}_calib_$c -o x --output-format csv -- tools/traffic_calib > gpurun_out/${T}_calib_$c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/*_calib_*/*counter_collection.csv") + glob.glob("gpurun_out/*_calib_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in acc.items():
    print(k, v)
PY
python3 tools/kstats.py gpurun_out/${T}_kernel_stats.csv | head -8
cut -c1-300 gpurun_out/${T}_bench.json
