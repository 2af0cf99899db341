cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 0 11 12 13; do
  export GAVISUNK_DEBUG_MODE=$m
  timeout -k 10 90 rocprofv3 --kernel-trace --stats -d gpurun_out/jm_$m -o x --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/jm_$m.log 2>&1 || { echo "mode $m failed"; exit 1; }
  echo "mode $m: $(python3 tools/kstats.py gpurun_out/jm_$m/x_kernel_stats.csv | grep -m3 k_join | tr '\n' ';' | tr -s ' ')"
done
