cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"]/1e9,1), round(d["ms_per_step"],3), d["roofline"].get("kernel_ms"))'; }
for t in 1024 768 512 256; do echo "probe threads $t: $(GAVISUNK_JOIN_THREADS=$t run)"; done
