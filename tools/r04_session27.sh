#!/bin/bash
# round 4, session 27: writer thread of the command line, parallel sidecar reads -- the file-facing tests, then the default bench
mkdir -p gpurun_out/r04
timeout -k 10 700 python3 -m pytest tests/test_tag_gpu.py tests/test_stream_gpu.py tests/test_pipeline_gpu.py tests/test_sidecar_gpu.py tests/test_ingest_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q --durations=5 > gpurun_out/r04/s27_tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r04/s27_tests.log
timeout -k 10 600 python3 bench.py > gpurun_out/r04/s27_bench.json 2> gpurun_out/r04/s27_bench.log
echo "bench rc=$?"; grep -E "bench \+" gpurun_out/r04/s27_bench.log | tail -8
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/s27_bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"frac",d["roofline"]["frac"],"traffic",d["roofline"]["traffic"])
c=d.get("cli_1e8_keys",{})
print({k:c.get(k) for k in ("parse_and_build_every_run","prepare_once","with_sidecar","same_bytes")})
g=d.get("cli_genome_scale",{})
print(g.get("seconds"), g.get("cli_bases_per_s"), json.dumps(g.get("seconds_by_phase")), g.get("head_of_output_equals_a_run_over_the_head_of_the_file"))
for k,v in d["config2"]["cli"].items():
    if isinstance(v,dict): print(k, v.get("seconds"), v.get("cli_bases_per_s"), v.get("bit_exact"))
PY
