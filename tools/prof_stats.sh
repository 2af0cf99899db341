# usage: bash tools/prof_stats.sh <tag> [bench args]   -> gpurun_out/r02/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/r02
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/prof_$tag -o x --output-format csv -- python3 bench.py --no-cpu-baseline --no-config2 "$@" > gpurun_out/r02/prof_$tag.json 2> gpurun_out/r02/prof_$tag.log || { tail -5 gpurun_out/r02/prof_$tag.log; exit 1; }
f=$(ls gpurun_out/r02/prof_$tag/*/x_kernel_stats.csv gpurun_out/r02/prof_$tag/x_kernel_stats.csv 2>/dev/null | head -1)
cp $f gpurun_out/r02/${tag}_kernel_stats.csv
python3 tools/kstats.py gpurun_out/r02/${tag}_kernel_stats.csv | head -14
rm -rf gpurun_out/r02/prof_$tag
