# usage: bash tools/rehearsal_k.sh <k> [more bench args]   BASELINE.json configs[4] rehearsal: 200 Mb, ~1e8 keys, 30 000 reads
mkdir -p gpurun_out/r02
k=$1; shift
timeout -k 10 500 python bench.py --k $k --genome-mb 200 --reads 30000 --steps 10 --warmup 2 --no-config2 --check-reads 200 --cpu-seconds 2 "$@" > gpurun_out/r02/c_k${k}_1e8keys_bench.json 2> gpurun_out/r02/c_k${k}.log || { tail -5 gpurun_out/r02/c_k${k}.log; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/c_k${k}_1e8keys_bench.json")); c=d["config"]
print("k=$k", round(d["value"]/1e9,1), "Gbase/s step ms", round(d["ms_per_step"],2), "frac", d["roofline"]["frac"], "kernel ms", d["roofline"]["kernel_ms"], "fallback", c["fallback_tiles"], "exact", d.get("parity",{}).get("bit_exact"), c["table_layout"])
PY
