#!/bin/bash
# round 4, session 30: the whole GPU suite, the gzip reader alone by thread count, the default bench
mkdir -p gpurun_out/r04
( time timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r04/s30_tests.log 2>&1
echo "tests rc=$?"; tail -22 gpurun_out/r04/s30_tests.log
GZ_TIME_THREADS=16,32,32 timeout -k 10 300 python3 tools/gz_time.py 6 > gpurun_out/r04/s30_gz_time.log 2>&1
tail -4 gpurun_out/r04/s30_gz_time.log
timeout -k 10 600 python3 bench.py > gpurun_out/r04/s30_bench.json 2> gpurun_out/r04/s30_bench.log
echo "bench rc=$?"; grep -E "bench \+" gpurun_out/r04/s30_bench.log | grep -E "steps:|pair stage|genome-scale" 
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/s30_bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"frac",d["roofline"]["frac"],"traffic",d["roofline"]["traffic"], "parity", d["parity"]["bit_exact"])
c=d.get("cli_1e8_keys",{})
print({k:c.get(k) for k in ("parse_and_build_every_run","prepare_once","with_sidecar","same_bytes")})
g=d.get("cli_genome_scale",{})
print(g.get("seconds"), g.get("cli_bases_per_s"), json.dumps(g.get("seconds_by_phase")), g.get("head_of_output_equals_a_run_over_the_head_of_the_file"))
for k,v in d["config2"]["cli"].items():
    if isinstance(v,dict): print(k, v.get("seconds"), v.get("cli_bases_per_s"), v.get("bit_exact"))
print("config2", d["config2"]["value"], d["config2"]["roofline"]["frac"], d["config2"]["pair_stage"]["parity"])
print("diploid", d["diploid_shard"]["value"], d["diploid_shard"]["roofline"]["frac"], d["diploid_shard"]["roofline"]["traffic"])
print("engine", d.get("headline_per_survey_8d"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
