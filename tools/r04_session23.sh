#!/bin/bash
# round 4, session 23: the device join of <db> x <loc> (gs_table_create_from_db) -- tests that load files, then the default bench
mkdir -p gpurun_out/r04
true
echo "tests rc=$?"; tail -12 gpurun_out/r04/s23_tests.log
timeout -k 10 600 python3 bench.py > gpurun_out/r04/s23_bench.json 2> gpurun_out/r04/s23_bench.log
echo "bench rc=$?"; grep -E "sunkdb|CLI at|bench \+" gpurun_out/r04/s23_bench.log | tail -30
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/s23_bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"frac",d["roofline"]["frac"],"traffic",d["roofline"]["traffic"])
print(json.dumps(d.get("cli_1e8_keys"),indent=1)[:1500])
print(json.dumps(d.get("cli_genome_scale",{}).get("seconds_by_phase")), d.get("cli_genome_scale",{}).get("seconds"))
print(json.dumps(d["config2"]["cli"],indent=1)[:1200])
PY
