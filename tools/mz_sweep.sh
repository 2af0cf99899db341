# usage: bash tools/mz_sweep.sh m:load [m:load ...]   minimiser length / keys-per-region sweep on the 100 Mb rehearsal of config 3
mkdir -p gpurun_out/r02
export GAVISUNK_BENCH_ALLOW_KNOBS=1
for cfg in "$@"; do
  m=${cfg%:*}; l=${cfg#*:}
  GAVISUNK_MZ_M=$m GAVISUNK_MZ_LOAD=$l timeout -k 10 120 python bench.py --genome-mb 100 --reads 20000 --steps 3 --warmup 1 --no-cpu-baseline --no-config2 > gpurun_out/r02/sw_${m}_$l.json 2> gpurun_out/r02/sw_${m}_$l.log
  python -c "
import json; d=json.load(open('gpurun_out/r02/sw_${m}_$l.json')); c=d['config']; print('m=$m load=$l', round(d['value']/1e9,1), 'Gbase/s', 'kernel_ms', d['roofline']['kernel_ms'], 'fallback', c['fallback_tiles'], '/', c['tiles'], 'table GiB', c['table_GiB'], c['table_layout'][-75:])"
done
