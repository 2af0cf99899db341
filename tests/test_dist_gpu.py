"""The read-sharded protocol with the REAL kernels: 2 and 3 processes share the one GPU of the test box (collectives
over gloo, since RCCL refuses two ranks on one device), each tags its shard with gs_tag_phase1 / phase2, and rank 0's
gathered hit list must equal a single-process pass over the whole input."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _gloo_helpers as gloo
from oracle import sunkdef

pytestmark = pytest.mark.gpu


def _case():
    from gavisunk_amd import synth

    k = 20
    rng = np.random.default_rng(77)
    g = rng.integers(0, 4, 60_000, dtype=np.uint8)
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], k)
    loc = (sunks.start // 900).astype(np.uint32)
    lens = synth.lognormal_lengths(rng, 53, median=2500, sigma=0.8, lo=1, hi=30000)
    reads = [synth.codes_to_ascii(r) for r in synth.simulate_reads(rng, [g], lens, sub=0.005, dele=0.005, ins=0.005)]
    reads[20] = b""  # shards that start or end on reads without raw hits
    reads[21] = b"ACGT" * 3
    return k, sunks.code, loc, reads


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gavisunk_amd import _lib, engine
        from gavisunk_amd import dist as gdist

        L = _lib.lib()
        k, keys, loc, reads = _case()
        tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
        lo, hi = gdist.shard_bounds(len(reads), world, rank)
        dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads[lo:hi]), k)
        has, last = C.c_uint32(0), C.c_uint32(0)
        _lib.check(L.gs_tag_phase1(tab._h, dev._h, C.byref(has), C.byref(last)))
        summ = gloo.exchange_summaries(has.value, last.value)
        _lib.check(L.gs_tag_phase2(tab._h, dev._h, gdist.resolve_carry(summ, rank)))
        n = C.c_uint64(0)
        _lib.check(L.gs_tag_finish(tab._h, dev._h, None, 0, C.byref(n), None, None))
        hits = np.zeros(max(n.value, 1), dtype=engine.HIT_DTYPE)
        _lib.check(L.gs_tag_hits_to_host(tab._h, dev._h, hits.ctypes.data, len(hits), None))
        hits = hits[: n.value]
        local = torch.from_numpy(np.stack([hits["read"], hits["pos"], hits["payload"]], axis=1).astype(np.int64).astype(np.int32)).reshape(-1, 3)
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
def test_sharded_gpu_run_equals_single_pass(world, tmp_path):
    from gavisunk_amd import engine

    out = str(tmp_path / "hits.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    k, keys, loc, reads = _case()
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    st, whole, _, _ = engine.tag(tab, engine.DeviceReads(engine.ReadBatch.from_ascii(reads), k))
    assert st == 0 and len(whole) > 50
    assert np.array_equal(got[:, 0], whole["read"].astype(np.int32))
    assert np.array_equal(got[:, 1], whole["pos"].astype(np.int32))
    assert np.array_equal(got[:, 2], whole["payload"].astype(np.int32))


def _comm_worker(rank, world, port, out_path, use_rccl):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gavisunk_amd import _lib, engine
    from gavisunk_amd import dist as gdist

    L = _lib.lib()
    device = rank % L.gs_device_count() if use_rccl else 0
    comm = gdist.Comm(device, rccl=use_rccl)
    k, keys, loc, reads = _case()
    tab = None
    if rank == 0 or not use_rccl:
        tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32), device)
    if use_rccl:
        tab = comm.rccl.broadcast_table(tab, root=0)  # the built table, device to device
    lo, hi = gdist.shard_bounds(len(reads), world, rank)
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads[lo:hi]), k, device)
    has, last = C.c_uint32(0), C.c_uint32(0)
    _lib.check(L.gs_tag_phase1(tab._h, dev._h, C.byref(has), C.byref(last)))
    summ = comm.exchange_summaries(has.value, last.value)
    _lib.check(L.gs_tag_phase2(tab._h, dev._h, gdist.resolve_carry(summ, rank)))
    n = C.c_uint64(0)
    _lib.check(L.gs_tag_finish(tab._h, dev._h, None, 0, C.byref(n), None, None))
    allhits = comm.gather_hits(tab, dev, n.value, lo)
    if rank == 0:
        np.save(out_path, allhits)
    dev.close()
    tab.close()
    comm.close()


def _check_against_single_pass(out):
    from gavisunk_amd import engine

    got = np.load(out)
    k, keys, loc, reads = _case()
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    st, whole, _, _ = engine.tag(tab, engine.DeviceReads(engine.ReadBatch.from_ascii(reads), k))
    assert st == 0 and len(whole) > 50 and np.array_equal(got, whole)


@pytest.mark.parametrize("world", [2, 3])
def test_comm_gloo_data_plane_on_one_gpu(world, tmp_path):
    """dist.Comm with the host-staged gather (what a run falls back to when RCCL cannot start): real kernels, shared GPU."""
    out = str(tmp_path / "hits.npy")
    mp.spawn(_comm_worker, args=(world, _free_port(), out, False), nprocs=world, join=True)
    _check_against_single_pass(out)


def _keyerror_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gavisunk_amd import _lib, engine
    from gavisunk_amd import dist as gdist
    from oracle import oracle

    L = _lib.lib()
    comm = gdist.Comm(0, rccl=False)
    k, keys, loc, reads = _case()
    # a db k-mer without loc row (kmerpos_annot3.nim:90 -> KeyError), met by ONE read in the middle of the file
    poison = b"ACGTTGCATTGACCAGTAGCATCAGGA"
    keys = np.append(keys, oracle.canonical_windows(poison, k)[:1])
    loc = np.append(loc, np.uint32(_lib.GS_MISSING))
    reads = reads[:26] + [reads[26] + poison + reads[27][:100]] + reads[27:]
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    lo, hi = gdist.shard_bounds(len(reads), world, rank)
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads[lo:hi]), k)
    has, last = C.c_uint32(0), C.c_uint32(0)
    _lib.check(L.gs_tag_phase1(tab._h, dev._h, C.byref(has), C.byref(last)))
    summ = comm.exchange_summaries(has.value, last.value)
    _lib.check(L.gs_tag_phase2(tab._h, dev._h, gdist.resolve_carry(summ, rank)))
    n = C.c_uint64(0)
    st = L.gs_tag_finish(tab._h, dev._h, None, 0, C.byref(n), None, None)
    code = 0
    try:
        comm.gather_hits(tab, dev, n.value, lo)
    except _lib.GsError as e:
        code = e.status
    np.save(f"{out_path}.{rank}.npy", np.array([st, code, int(lo <= 26 < hi)]))
    dev.close()
    tab.close()
    comm.close()


def test_a_keyerror_on_one_shard_fails_every_rank(tmp_path):
    """The reference exits 1 when a read hits a db k-mer that has no loc row (kmerpos_annot3.nim:90).  With the reads sharded
    that happens on ONE rank; the gather carries every rank's status with its count, so all ranks report the KeyError instead
    of waiting for the failed one (3 processes sharing the GPU, gloo data plane: the same protocol gs_gather_hits follows
    over RCCL)."""
    out = str(tmp_path / "st")
    mp.spawn(_keyerror_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    from gavisunk_amd import _lib

    res = [np.load(f"{out}.{r}.npy") for r in range(3)]
    assert sum(int(x[2]) for x in res) == 1
    for x in res:
        assert x[1] == _lib.GS_E_KEYERROR                      # every rank learns of it
        assert (x[0] == _lib.GS_E_KEYERROR) == bool(x[2])       # only the shard with the read met it itself


def test_comm_rccl_data_plane():
    """The RCCL data plane (table broadcast + hit gather through the C-ABI) on as many GPUs as the box has, at most 4;
    skipped on a one-GPU box (RCCL refuses two ranks on one device)."""
    import tempfile

    from gavisunk_amd import _lib

    ndev = _lib.lib().gs_device_count()
    if ndev < 2:
        pytest.skip("needs >= 2 GPUs")
    world = min(ndev, 4)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "hits.npy")
        mp.spawn(_comm_worker, args=(world, _free_port(), out, True), nprocs=world, join=True)
        _check_against_single_pass(out)


def test_rccl_single_rank_through_the_c_abi():
    """gs_comm_* with one rank (all a one-GPU box can run): communicator, table broadcast (the root gets its own table back),
    gather = the rank's own list with the read offset applied."""
    from gavisunk_amd import _lib, engine, rccl

    class Solo:
        rank, world = 0, 1

        @staticmethod
        def broadcast_bytes(b, src=0):
            return bytes(b)

    k, keys, loc, reads = _case()
    comm = rccl.RcclComm(Solo, 0)
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    assert comm.broadcast_table(tab) is tab
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads), k)
    st, whole, _, _ = engine.tag(tab, dev)
    got = comm.gather_hits(tab, dev, [len(whole)], 1000)
    assert len(got) == len(whole) > 50 and np.array_equal(got["read"], whole["read"] + 1000)
    assert np.array_equal(got["pos"], whole["pos"]) and np.array_equal(got["payload"], whole["payload"])
    # the rows of a finished pass stay valid in device memory while the NEXT pass's probe kernels are queued (gs_tag_phase1_launch
    # returns at once): what lets bench.py run the gather of pass i beside the probing of pass i + 1
    import ctypes as C

    L = _lib.lib()
    _lib.check(L.gs_tag_phase1_launch(tab._h, dev._h))
    again = comm.gather_hits(tab, dev, [len(whole)], 7)
    assert np.array_equal(again["read"], whole["read"] + 7) and np.array_equal(again["pos"], whole["pos"]) and np.array_equal(again["payload"], whole["payload"])
    has, last = C.c_uint32(0), C.c_uint32(0)
    _lib.check(L.gs_tag_phase1_wait(tab._h, dev._h, C.byref(has), C.byref(last)))
    assert has.value == 1
    _lib.check(L.gs_tag_phase2(tab._h, dev._h, _lib.GS_NONE))
    n = C.c_uint64(0)
    _lib.check(L.gs_tag_finish(tab._h, dev._h, None, 0, C.byref(n), None, None))
    assert n.value == len(whole)
    got = comm.gather_hits(tab, dev, [len(whole)], 0)
    assert np.array_equal(got, whole)
    # the phase-1 summaries over ncclAllGather (gs_comm_allgather_u64), and a failed pass reported by the gather itself
    assert comm.allgather_pairs(1, 0xFFFFFFFF) == [(1, 0xFFFFFFFF)]
    keys2, loc2 = np.append(keys, keys[:1] ^ np.uint64(1)), np.append(loc, np.uint32(_lib.GS_MISSING))
    from oracle import oracle

    bad = engine.DeviceTable(k, np.append(keys, oracle.canonical_windows(b"ACGTTGCATTGACCAGTAGCATCAGGA", k)[:1]), loc2, np.arange(len(keys2), dtype=np.uint32))
    dev2 = engine.DeviceReads(engine.ReadBatch.from_ascii(reads[:5] + [b"ACGTTGCATTGACCAGTAGCATCAGGA"]), k)
    assert engine.tag(bad, dev2)[0] == _lib.GS_E_KEYERROR
    with pytest.raises(_lib.GsError) as e:
        comm.gather_hits(bad, dev2, None, 0)
    assert e.value.status == _lib.GS_E_KEYERROR
    comm.close()
