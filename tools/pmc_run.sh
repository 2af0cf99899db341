# usage: bash tools/pmc_run.sh <tag> "<counters>" ["<counters>" ...]   (one rocprofv3 --pmc pass per counter group)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
i=0
for grp in "$@"; do
  rocprofv3 --pmc $grp -d gpurun_out/pmc_${tag}_$i -o x --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_$i.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_$i.log; exit 1; }
  i=$((i+1))
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* | tee gpurun_out/pmc_${tag}.txt
