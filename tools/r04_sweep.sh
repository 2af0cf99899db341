#!/bin/bash
# round 4: surrogate realism sweep -- config 3 with location groups of ~200 / ~2000 SUNKs (the default surrogate: ~12 000): kernel
# time, frac, table size, overflow keys (bench line) and what the region fetches find (tuning library's census)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
for gl in 200 2000; do
  timeout -k 10 500 python3 bench.py --group-len $gl --steps 4 --warmup 1 --no-cpu-baseline --no-config2 --no-engine --no-chain > $o/sweep_gl$gl.json 2> $o/sweep_gl$gl.log || { tail -5 $o/sweep_gl$gl.log; exit 1; }
  GAVISUNK_BENCH_ALLOW_KNOBS=1 GAVISUNK_LIB=$PWD/gavisunk_amd/libgavisunk_tuning.so timeout -k 10 500 python3 bench.py --group-len $gl --steps 1 --warmup 0 --no-cpu-baseline --no-config2 --no-engine --no-chain > $o/sweep_gl${gl}_census.json 2> $o/sweep_gl${gl}_census.log || { tail -5 $o/sweep_gl${gl}_census.log; exit 1; }
  python3 - $o/sweep_gl$gl.json $o/sweep_gl${gl}_census.json $gl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[0]); r = d["roofline"]; c = d["config"]
z = json.loads(open(sys.argv[2]).read().strip().splitlines()[0])["config"].get("census_of_one_pass", {})
w = max(z.get("windows", 1), 1)
print(f"group-len {sys.argv[3]:>5s}: {c['sunks']} SUNKs in {c['sunk_groups']} groups ({c['sunks'] / max(c['sunk_groups'], 1):.0f} per group), table {c['table_GiB']} GiB | kernel {r['kernel_ms_per_pass']:.1f} ms/pass frac {r['frac']:.4f} step {d['value'] / 1e9:.1f} Gbase/s h_raw {c['h_raw']} | "
      f"runs/window {z.get('runs', 0) / w:.4f} deferred/window {z.get('windows_deferred', 0) / w:.4f} hot-region runs {z.get('runs_hot_region', 0) / max(z.get('runs', 1), 1):.3f} | {c['table_layout'][:100]}")
PY
done
