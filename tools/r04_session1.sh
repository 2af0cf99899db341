#!/bin/bash
# round 4, GPU session 1: self-launch rehearsal, census of the region fetches (tuning library), cache-policy A/B of the region DMA,
# PMC of the diploid shard
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
timeout -k 10 900 python3 -m pytest tests/test_bench_selflaunch.py -m gpu -x -q > $o/s1_selflaunch.log 2>&1; echo "selflaunch rc=$?"; tail -3 $o/s1_selflaunch.log
timeout -k 10 600 python3 -m pytest tests/test_fuzz_gpu.py -m gpu -x -q --durations=5 > $o/s1_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -12 $o/s1_fuzz.log
GAVISUNK_BENCH_ALLOW_KNOBS=1 GAVISUNK_LIB=$PWD/gavisunk_amd/libgavisunk_tuning.so timeout -k 10 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-config2 --no-engine > $o/s1_census.json 2> $o/s1_census.log || { tail -5 $o/s1_census.log; exit 1; }
grep census $o/s1_census.log
for v in base nt sc1; do
  lib=$PWD/gavisunk_amd/libgavisunk_$v.so; [ "$v" = base ] && lib=$PWD/gavisunk_amd/libgavisunk.so
  GAVISUNK_BENCH_ALLOW_KNOBS=1 GAVISUNK_LIB=$lib timeout -k 10 600 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config2 --no-engine > $o/s1_ab_$v.json 2> $o/s1_ab_$v.log || { tail -5 $o/s1_ab_$v.log; exit 1; }
  python3 - $o/s1_ab_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; e = d.get("diploid_shard", {})
print(f"{sys.argv[2]:6s} config3: kernel {r['kernel_ms_per_pass']:.2f} ms/pass frac {r['frac']:.4f} step {d['value']/1e9:.1f} Gbase/s | diploid: kernel {e.get('roofline',{}).get('kernel_ms_per_pass',0):.2f} ms frac {e.get('roofline',{}).get('frac',0):.4f} step {e.get('value',0)/1e9:.1f}")
PY
done
tools/pmc.sh r04 dip0 --workload diploid > /dev/null 2>&1; cat $o/dip0_pmc.txt
