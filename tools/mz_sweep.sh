mkdir -p gpurun_out/r02
export GAVISUNK_BENCH_ALLOW_KNOBS=1
for cfg in "14 4" "15 4" "16 4" "14 2" "15 2" "16 2" "16 3"; do
  set -- $cfg
  GAVISUNK_MZ_M=$1 GAVISUNK_MZ_LOAD=$2 timeout -k 10 120 python bench.py --genome-mb 100 --reads 20000 --steps 3 --warmup 1 --no-cpu-baseline --no-config2 > gpurun_out/r02/sw_$1_$2.json 2> gpurun_out/r02/sw_$1_$2.log
  python -c "
import json; d=json.load(open('gpurun_out/r02/sw_$1_$2.json')); c=d['config']; print('m=$1 load=$2', round(d['value']/1e9,1), 'Gbase/s', 'kernel_ms', d['roofline']['kernel_ms'], 'fallback', c['fallback_tiles'], '/', c['tiles'], c['table_layout'][-70:])"
done
