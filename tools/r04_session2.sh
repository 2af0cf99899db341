#!/bin/bash
# round 4, GPU session 2: the device chain's tests, the table-size dependence of the probe kernel's miss rate, a full bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_chain_gpu.py tests/test_diag_gpu.py tests/test_pair_golden.py tests/test_pair_gpu.py tests/test_pipeline_gpu.py tests/test_badsunks_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q --durations=8 > $o/s2_tests.log 2>&1; echo "tests rc=$?"; tail -25 $o/s2_tests.log
for mb in 200 800; do
  GAVISUNK_BENCH_ALLOW_KNOBS=1 GAVISUNK_MZ_M=15 timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --genome-mb $mb --reads 300000 --no-cpu-baseline --no-config2 --no-engine --no-extra --no-chain > $o/s2_size_$mb.json 2> $o/s2_size_$mb.log || { tail -5 $o/s2_size_$mb.log; }
  python3 - $o/s2_size_$mb.json $mb <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; c = d["config"]
print(f"genome {sys.argv[2]} Mb: table {c['table_GiB']} GiB {c['table_layout'][:60]} | kernel {r['kernel_ms_per_pass']:.2f} ms/pass, {c['windows'] / r['kernel_ms_per_pass'] / 1e6:.1f} G windows/s, h_raw {c['h_raw']}")
PY
done
timeout -k 10 900 python3 bench.py > $o/s2_bench.json 2> $o/s2_bench.log || { tail -20 $o/s2_bench.log; }
grep -h "Gbase\|pair stage" $o/s2_bench.log
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04/s2_bench.json"))
print(json.dumps(d.get("pair_stage"), indent=1)[:3000])
print(json.dumps(d.get("config2", {}).get("pair_stage"), indent=1)[:3000])
print(d.get("headline_per_survey_8d"))
PY
