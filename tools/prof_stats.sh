# usage: bash tools/prof_stats.sh <tag> [bench args]   -> gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o x --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1 || exit 1
f=$(ls gpurun_out/prof_$tag/*/x_kernel_stats.csv gpurun_out/prof_$tag/x_kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
cut -c1-150 gpurun_out/${tag}_kernel_stats.csv | head -12
