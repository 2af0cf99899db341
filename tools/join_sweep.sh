cd $GRAFT_REPO_ROOT
export GAVISUNK_TABLE=compact-lds
for cfg in "$@"; do
  set -- $cfg
  echo "pipe $1 reps $2: $(GAVISUNK_JOIN_PIPE=$1 GAVISUNK_JOIN_REPS=$2 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["value"]/1e9,1), round(d["ms_per_step"],3), d["roofline"].get("kernel_ms"))')"
done
