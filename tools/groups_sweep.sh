cd $GRAFT_REPO_ROOT
for g in 0 20000 2000 700; do
  for lay in auto compact-l2; do
    if [ $lay = auto ]; then export GAVISUNK_TABLE=compact-lds; else export GAVISUNK_TABLE=$lay; fi
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --split-groups $g 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; c=d["config"]; print(sys.argv[1], sys.argv[2], round(d["value"]/1e9,1), "Gbase/s kernel", r["kernel_ms"], "emitted", c["emitted_rows"], c["table_layout"][:140])' $g $lay
  done
done
