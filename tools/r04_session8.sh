#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
timeout -k 10 600 python3 -m pytest tests/test_tag_gpu.py tests/test_ingest_gpu.py tests/test_fullsize_gpu.py tests/test_sidecar_gpu.py -m gpu -x -q > $o/s8_tests.log 2>&1; echo "tests rc=$?"; tail -3 $o/s8_tests.log
( time timeout -k 10 900 python3 bench.py --steps 3 --warmup 1 --no-extra --no-engine --no-chain --no-cpu-baseline ) > $o/s8_bench.json 2> $o/s8_bench.log || { tail -20 $o/s8_bench.log; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/s8_bench.json").read().strip().splitlines()[0])
for k, v in d["config2"]["cli"].items():
    if isinstance(v, dict): print(k, v["seconds"], round(v["cli_bases_per_s"] / 1e9, 3), v["bit_exact"], v["seconds_by_phase"])
g = d.get("cli_genome_scale"); print("cli_genome_scale", g and (g["seconds"], g["cli_bases_per_s"] / 1e9, g["seconds_by_phase"], g["head_of_output_equals_a_run_over_the_head_of_the_file"]))
print("cli_1e8", {k: v for k, v in d.get("cli_1e8_keys", {}).items() if k != "note"})
PY
