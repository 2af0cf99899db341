"""`python bench.py --gpus N` (N > 1, no launcher): bench.self_launch starts the ranks as child processes through
torch.distributed.run, relays rank 0's JSON line and passes the exit code on.  CPU: a stand-in rank program on gloo;
GPU: the real bench.py, two ranks sharing the box's one device over the gloo data plane (reduced workload)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "_selflaunch_rank.py")


def _run(args):
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(2, %r, %r))" % (ROOT, STANDIN, args))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)


def test_self_launch_relays_one_json_line():
    p = _run(["--x", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d == {"metric": "selftest", "n_gpus": 2, "sum": 3, "args": ["--x", "1"]}
    assert "some library chatter" in p.stderr  # (everything else the ranks print on stdout goes to stderr)


def test_self_launch_passes_failure_on():
    p = _run(["--fail"])
    assert p.returncode != 0  # (a line rank 0 managed to print is still relayed; the exit code says the run failed)


@pytest.mark.gpu
def test_bench_gpus2_runs_by_itself_on_one_device():
    env = dict(os.environ, GAVISUNK_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-mb", "40",
                        "--reads", "3000", "--check-reads", "200", "--cpu-seconds", "1", "--rehearse-extra", "--sharded-chain"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["parity"]["bit_exact"] is True and d["config"]["gathered_list_checked"] is True
    assert "gloo" in d["config"]["data_plane"]
    # the extra key of an N > 1 line: BASELINE.json configs[3] cut over the ranks (strong scaling), with its no-exchange reference
    e = d["diploid_strong"]
    assert e["scaling"] == "strong" and e["value"] > 0 and e["config"]["read_shards_per_gpu"] == 4 and e["no_exchange"]["value"] > 0
    # --sharded-chain: the second half over both ranks' own hit lists, merged on rank 0
    ps = d["pair_stage"]
    assert ps["ranks"] == 2 and ps["rows_in"] > 10000 and ps["rows_after_diag_filter"] > 1000 and ps["tsv_rows"] > 100 and ps["validated_intervals"] > 0
