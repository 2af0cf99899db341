"""World-size-2 (and 3) `gloo` test of the read-sharded multi-GPU protocol (gavisunk_amd/dist.py) on CPU.

There is no CPU tagging path in the product, so here the per-shard "device" work is played by
the oracle; what is under test is the host logic that makes N shards equal one file: shard
bounds, the carry exchange (all_gather of {has_raw, last_loc}), carry resolution, the gather of
hit buffers in shard order with read-index offsets, and the table broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _gloo_helpers as gloo

from gavisunk_amd import dist as gdist
from gavisunk_amd import synth
from gavisunk_amd._lib import GS_NONE
from oracle import oracle, sunkdef


def _case():
    k = 20
    rng = np.random.default_rng(11)
    g = rng.integers(0, 4, 40_000, dtype=np.uint8)
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], k)
    loc = (sunks.start // 700).astype(np.uint32)  # coarse groups: carries across reads do happen
    lens = synth.lognormal_lengths(rng, 41, median=900, sigma=0.7, lo=1, hi=20000)
    reads = [synth.codes_to_ascii(r) for r in synth.simulate_reads(rng, [g], lens, sub=0.005, dele=0.005, ins=0.005)]
    return k, sunks.code, loc, reads


def _oracle_shard(k, keys, loc, seqs, carry_loc):
    """Oracle stand-in for gs_tag_phase1/phase2 on one shard; locs play the role of loc rows."""
    t = oracle.Table(k, keys, keys, np.zeros(len(keys), np.uint32), loc.astype(np.int64))
    buf, off = oracle.concat_reads(seqs)
    prev = oracle.NONE
    if carry_loc != GS_NONE:
        prev = int(np.nonzero(loc == carry_loc)[0][0])  # any row of that loc
    rc, hits, prev_out, ctr = t.tag(buf, off, prev_row=prev)
    assert rc == 0
    last_loc = int(loc[prev_out]) if prev_out != oracle.NONE else GS_NONE
    has_raw = int(ctr["raw_hits"] > 0)
    return hits, has_raw, (last_loc if has_raw else GS_NONE)


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, keys, loc, reads = _case()
        # table arrays travel from rank 0 through the collective backend
        keys_b = gloo.broadcast_array(keys if rank == 0 else None, np.uint64)
        loc_b = gloo.broadcast_array(loc if rank == 0 else None, np.uint32)
        assert np.array_equal(keys_b, keys) and np.array_equal(loc_b, loc)
        lo, hi = gdist.shard_bounds(len(reads), world, rank)
        shard = reads[lo:hi]
        # phase 1 (no carry known yet): summary of the shard
        _, has_raw, last_loc = _oracle_shard(k, keys_b, loc_b, shard, GS_NONE)
        summ = gloo.exchange_summaries(has_raw, last_loc)
        carry = gdist.resolve_carry(summ, rank)
        # phase 2 with the resolved carry
        hits, _, _ = _oracle_shard(k, keys_b, loc_b, shard, carry)
        local = torch.from_numpy(np.stack([hits["read"], hits["pos"], hits["row"]], axis=1).astype(np.int32)).reshape(-1, 3)
        allhits = gloo.gather_hits(local, lo)
        if rank == 0:
            np.save(out_path, allhits.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_equals_single_file(world, tmp_path):
    out = str(tmp_path / "hits.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    k, keys, loc, reads = _case()
    whole, _, _ = _oracle_shard(k, keys, loc, reads, GS_NONE)
    assert len(whole) > 20
    assert np.array_equal(got[:, 0], whole["read"].astype(np.int32))
    assert np.array_equal(got[:, 1], whole["pos"].astype(np.int32))
    # same loc per row (rows of equal loc are interchangeable for the stand-in carry)
    assert np.array_equal(loc[got[:, 2]], loc[whole["row"]])


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            b = [gdist.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_resolve_carry():
    s = [(0, GS_NONE), (1, 5), (0, GS_NONE), (1, 9)]
    assert gdist.resolve_carry(s, 0) == GS_NONE
    assert gdist.resolve_carry(s, 0, carry0=3) == 3
    assert gdist.resolve_carry(s, 1) == GS_NONE
    assert gdist.resolve_carry(s, 2) == 5
    assert gdist.resolve_carry(s, 3) == 5


def _ctrl_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = gdist.Comm(device=0, rccl=False)  # control plane only: no GPU here
    summ = comm.exchange_summaries(rank % 2, 100 + rank)
    mx = comm.all_max(1.5 * (rank + 1))
    tot = comm.all_sum(10 ** 12 + rank)
    blob = comm.broadcast_bytes(bytes(range(128)) if rank == 0 else b"", src=0)
    comm.barrier()
    if rank == world - 1:
        np.save(out_path, np.array([summ == [(r % 2, 100 + r) for r in range(world)], mx == 1.5 * world,
                                    tot == world * 10 ** 12 + sum(range(world)), blob == bytes(range(128))]))
    comm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_comm_control_plane(world, tmp_path):
    """dist.Comm: what travels over gloo in a multi-GPU run (phase-1 summaries, scalars, the RCCL unique id's 128 bytes)."""
    out = str(tmp_path / "ok.npy")
    mp.spawn(_ctrl_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.load(out).all()


def _failing_worker(rank, world, port, out_dir):
    """Two passes: the first is fine; in the second, rank 1's phase 1 fails (as a GsError of a kernel launch or an allocation would).
    It still takes part in the exchange, with the status word; EVERY rank raises after it and exits non-zero."""
    import sys
    import time

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = gdist.Comm(device=0, rccl=False)
    t0 = time.time()
    summ = gdist.checked_summaries(comm, rank % 2, 7 + rank)
    assert summ == [(r % 2, 7 + r) for r in range(world)]
    try:
        gdist.checked_summaries(comm, 1, 3, error=RuntimeError("injected: phase 1 failed") if rank == 1 else None)
    except gdist.RankFailed as e:
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(f"{time.time() - t0:.3f}\n{e}\n")
        dist.destroy_process_group()
        sys.exit(3)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write("no error raised\n")


def test_a_failing_rank_takes_every_rank_down_in_bounded_time(tmp_path):
    """bench.py --gpus N / dist.py: the ONE exchange of a pass carries a status word; a rank whose pass failed before it poisons its
    summary instead of leaving, so no rank is left waiting inside a collective (gloo's default timeout is 30 minutes)."""
    import time

    world = 3
    t0 = time.time()
    ctx = mp.spawn(_failing_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    deadline = t0 + 120
    while any(p.is_alive() for p in ctx.processes) and time.time() < deadline:
        time.sleep(0.2)
    alive = [p.is_alive() for p in ctx.processes]
    for p in ctx.processes:
        if p.is_alive():
            p.kill()
    assert not any(alive), "a rank was still waiting two minutes after another one had failed"
    assert [p.exitcode for p in ctx.processes] == [3, 3, 3]
    for r in range(world):
        secs, msg = (tmp_path / f"rank{r}.txt").read_text().split("\n", 1)
        assert float(secs) < 60 and "rank(s) [1]" in msg
