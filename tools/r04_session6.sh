#!/bin/bash
# round 4, GPU session 6: GPU suite, the default bench line (traffic quoted for every workload), diploid with m = 14
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
( time timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q ) > $o/s6_tests.log 2>&1; echo "tests rc=$?"; tail -6 $o/s6_tests.log
( time timeout -k 10 900 python3 bench.py ) > $o/s6_bench.json 2> $o/s6_bench.log || { tail -20 $o/s6_bench.log; }
grep -h "Gbase\|pair stage\|genome-scale\|real" $o/s6_bench.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/s6_bench.json").read().strip().splitlines()[0])
print("value", d["value"] / 1e9, "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "| diploid traffic", d["diploid_shard"]["roofline"]["traffic"])
print("cli", json.dumps(d["config2"].get("cli"), indent=1))
g = d.get("cli_genome_scale"); print("cli_genome_scale", g and (g["seconds"], g["cli_bases_per_s"] / 1e9, g["seconds_by_phase"]))
print("cli_1e8", {k: v for k, v in d.get("cli_1e8_keys", {}).items() if k != "note"})
PY
GAVISUNK_BENCH_ALLOW_KNOBS=1 GAVISUNK_MZ_M=14 timeout -k 10 400 python3 bench.py --workload diploid --steps 4 --warmup 1 --no-cpu-baseline --no-config2 --no-engine --no-extra --no-chain > $o/s6_dip_m14.json 2> $o/s6_dip_m14.log || tail -5 $o/s6_dip_m14.log
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04/s6_dip_m14.json").read().strip().splitlines()[0]); r = d["roofline"]; c = d["config"]
    print("diploid m=14:", c["table_layout"][:160], "| table", c["table_GiB"], "GiB kernel", r["kernel_ms_per_pass"], "ms frac", r["frac"], "step", d["value"] / 1e9)
except Exception as e:
    print("diploid m=14 failed:", e)
PY
