#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline workload WITH the device-resident second half (`pair_stage`, `engine_both_halves`,
# `cli_config3_rows` legs): which kernels the chain is made of (gpurun_out/$ROUND/chain_config3_kernel_stats.csv -> profiles/$ROUND/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/${ROUND:-r05}; mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o/prof_chain -o x --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config2 --no-extra --no-engine > $o/chain_config3_bench_under_rocprof.json 2> $o/prof_chain.log || { tail -5 $o/prof_chain.log; exit 1; }
f=$(ls $o/prof_chain/*/x_kernel_stats.csv $o/prof_chain/x_kernel_stats.csv 2>/dev/null | head -1)
cp $f $o/chain_config3_kernel_stats.csv
rm -rf $o/prof_chain
grep "pair stage" $o/prof_chain.log
python3 - $o <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/chain_config3_kernel_stats.csv")))
keep = [r for r in rows if any(t in r["Name"] for t in ("k_rows_from", "k_seg", "k_tie", "k_scratch", "k_col_max", "k_mark_present", "k_shift", "k_gather6", "k_uf_", "k_slots", "k_diag", "k_keep", "k_count", "k_hist", "k_mark", "k_not_bad", "k_group", "k_flag", "k_dedup", "k_final", "k_out_rows", "k_vertex", "k_lookup", "k_region", "k_pair_", "k_uf_", "k_head", "k_pair_fill", "k_col_key", "k_gather", "k_iota", "k_dec", "k_any", "k_widen", "k_unique", "k_pack"))]
tot = sum(float(r["TotalDurationNs"]) for r in keep)
print(f"chain kernels (rocprim sorts / scans not counted): {tot / 1e6:.1f} ms")
sorts = [r for r in rows if 'rocprim' in r['Name']]
for r in sorted(sorts, key=lambda r: -float(r['TotalDurationNs']))[:8]:
    import re
    m = re.search(r'wrapped_(\w+?)_config', r['Name'])
    print(f'rocprim {m.group(1) if m else "?":40s} calls {r["Calls"]:>4s} total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
for r in sorted(keep, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
PY
