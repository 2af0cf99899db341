cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"]/1e9,1), round(d["ms_per_step"],3), d["roofline"].get("kernel_ms"))'; }
echo "l2: $(GAVISUNK_TABLE=compact-l2 run)"
export GAVISUNK_TABLE=compact-lds  # (the LDS join: the default before round 2)
for p in 3 4 5; do echo "lds pipe $p: $(GAVISUNK_JOIN_PIPE=$p run)"; done
export GAVISUNK_TABLE=compact-lds  # (the LDS join: the default before round 2)
echo "l2: $(GAVISUNK_TABLE=compact-l2 run)"
