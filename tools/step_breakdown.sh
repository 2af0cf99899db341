cd $GRAFT_REPO_ROOT
for i in 1 2; do python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("step", round(d["ms_per_step"],4), "pass", r["pass_ms"], "kernel", r["kernel_ms"], "-> gpu tail", round(r["pass_ms"]-r["kernel_ms"],4), "host", round(d["ms_per_step"]-r["pass_ms"],4))'; done
