#!/bin/bash
# round 4, GPU session 3: the whole GPU suite (timed), then the default bench line with every leg
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
( time timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=15 ) > $o/s3_tests.log 2>&1; echo "tests rc=$?"; tail -30 $o/s3_tests.log
( time timeout -k 10 900 python3 bench.py ) > $o/s3_bench.json 2> $o/s3_bench.log || { tail -20 $o/s3_bench.log; }
grep -h "Gbase\|pair stage\|genome-scale\|real" $o/s3_bench.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/s3_bench.json").read().strip().splitlines()[0])
print("value", d["value"] / 1e9, "frac", d["roofline"]["frac"])
print("cli", json.dumps(d["config2"].get("cli"), indent=1))
print("cli_genome_scale", json.dumps(d.get("cli_genome_scale"), indent=1))
print("cli_1e8", json.dumps(d.get("cli_1e8_keys"), indent=1)[:1500])
PY
