#!/bin/bash
# Kernel A/B on the GPU box: for every library variant given (names of gavisunk_amd/libgavisunk_<name>.so, "base" = the product
# library) the minimiser-layout parity tests, then a quick bench (an HBM-resident rehearsal of config 3 with its geometry, m = 17,
# and config 2 through the same kernel).  One line per variant in gpurun_out/r03/kbench.txt.
#   tools/kbench.sh base v1 v2 ...        (KBENCH_FULL=1: the full config-3 workload instead of the rehearsal)
mkdir -p gpurun_out/r03
out=gpurun_out/r03/kbench.txt
for v in "$@"; do
  lib=$PWD/gavisunk_amd/libgavisunk_$v.so
  [ "$v" = base ] && lib=$PWD/gavisunk_amd/libgavisunk.so
  export GAVISUNK_LIB=$lib
  t=ok
  if [ -z "$KBENCH_NOTEST" ]; then
    timeout -k 10 300 python -m pytest tests/test_tag_gpu.py -m gpu -q -x -k "auto or minimizer" > gpurun_out/r03/kb_$v.test.log 2>&1 || t=FAILED
    rc=$?
  fi
  if [ -n "$KBENCH_FULL" ]; then
    GAVISUNK_BENCH_ALLOW_KNOBS=1 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-extra --no-engine --no-cpu-baseline --no-cli > gpurun_out/r03/kb_$v.json 2> gpurun_out/r03/kb_$v.err || t="$t bench-failed"
  else
    GAVISUNK_BENCH_ALLOW_KNOBS=1 GAVISUNK_MZ_M=${KBENCH_M:-17} timeout -k 10 120 python bench.py --steps 6 --warmup 2 --genome-mb 400 --reads 150000 --no-extra --no-engine --no-cpu-baseline --no-cli > gpurun_out/r03/kb_$v.json 2> gpurun_out/r03/kb_$v.err || t="$t bench-failed"
  fi
  python - "$v" "$t" >> $out <<'PY'
import json, sys
v, t = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/r03/kb_{v}.json").read())
    r, c2 = d["roofline"], d.get("config2", {})
    print(f"{v:10s} tests {t:8s} | big: kernel {r['kernel_ms_per_pass']:8.3f} ms/pass  {r['kernel_bases_per_s'] / 1e9:6.1f} Gbase/s  frac {r['frac']:.4f}  step {d['value'] / 1e9:6.1f} | "
          f"config2: kernel {c2.get('roofline', {}).get('kernel_ms', 0):.4f} ms  frac {c2.get('roofline', {}).get('frac', 0):.4f}  parity {c2.get('parity', {}).get('bit_exact')}")
except Exception as e:
    print(f"{v:10s} tests {t} | no bench line: {e}")
PY
done
unset GAVISUNK_LIB
cat $out
