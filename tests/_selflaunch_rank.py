"""A stand-in rank program for tests/test_bench_selflaunch.py: what bench.py's ranks do on the control plane (gloo rendezvous,
barrier, all-reduce), rank 0 prints ONE JSON line on stdout, noise goes to stdout before it; `--fail` makes rank 1 exit 3."""
import json
import os
import sys

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
t = torch.tensor([rank + 1], dtype=torch.int64)
dist.all_reduce(t)
dist.barrier()
if "--fail" in sys.argv and rank == 1:
    dist.destroy_process_group()
    sys.exit(3)
if rank == 0:
    print("some library chatter on stdout")
    print(json.dumps({"metric": "selftest", "n_gpus": world, "sum": int(t.item()), "args": sys.argv[1:]}))
dist.destroy_process_group()
