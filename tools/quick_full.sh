# usage: bash tools/quick_full.sh <tag>   full-size config-3 bench (3 steps, small oracle check): one summary line
mkdir -p gpurun_out/r02
tag=$1; shift
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-config2 --check-reads 300 --cpu-seconds 2 "$@" > gpurun_out/r02/bench_$tag.json 2> gpurun_out/r02/bench_$tag.log || { tail -5 gpurun_out/r02/bench_$tag.log; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02/bench_$tag.json")); c=d["config"]
print("$tag", round(d["value"]/1e9,1), "Gbase/s step ms", round(d["ms_per_step"],1), "frac", d["roofline"]["frac"], "kernel ms", d["roofline"]["kernel_ms"], "fallback", c["fallback_tiles"], "raw", c["raw_hits"], "GiB", c["table_GiB"], "exact", d.get("parity",{}).get("bit_exact"))
PY
