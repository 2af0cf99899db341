"""The read-sharded SECOND half (gavisunk_amd.dist.sharded_second_half over csrc/gs_chain.hip; SURVEY.md section 8e: "pair-distance
stage: shard by read; the contig-wide union is a merge"): 2 and 3 processes share the test box's one GPU (exchanges over gloo:
RCCL refuses two ranks on one device), each runs diagonal filter -> step 2 -> bad SUNKs -> pair stage over the rows of its OWN
reads, the root merges the TSV rows and runs the union -- and everything must equal ONE process over all rows: the diagonal
filter's verdicts (the Table capacity carried across shard boundaries), the bad-SUNK flags (counters summed over ranks), the TSV
rows and the validated intervals.  The same exchanges through the C-ABI's RCCL entry points run with one rank."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _rows(seed, n_reads, big_at):
    """Rows of n_reads reads: 6 "main" contigs carry SUNKs every ~1.5 kb and the reads cover them ~6-fold along a diagonal (the
    bad-SUNK histogram has a mode, pairs of SUNKs pass the distance test, neighbouring reads share IDs for the union); 124 more
    contigs give stray hits.  A third of the reads are TIES for the diagonal filter: the same number of hits, all good, on two
    contigs -- the slot order of Nim's Table decides, i.e. the Table's capacity.  The reads in big_at hit 44-90 contigs: the
    Table grows there and stays grown for every later read, on whichever rank it lives."""
    rng = np.random.default_rng(seed)
    contigs = [f"ctg{j:03d}_{int(rng.integers(0, 10**5))}" for j in range(130)]
    n_main, clen = 6, 200_000
    sunk_at = [np.sort(rng.choice(np.arange(1000, clen, 50), size=clen // 1500, replace=False)) for _ in range(n_main)]
    name, pos, contig, start, ident, rlen = [], [], [], [], [], []
    for r in range(n_reads):
        per = []
        tie = r % 3 == 1 and r not in big_at
        c = int(rng.integers(0, n_main))
        ln = int(rng.choice([4000, 12000, 30000, 30000]))
        o = int(rng.integers(0, clen - ln))
        mine = sunk_at[c][(sunk_at[c] >= o) & (sunk_at[c] < o + ln)]
        if tie:
            mine = mine[: int(rng.integers(2, 4))]
        for s in mine.tolist():
            per.append((s - o + 20, c, s, s))
        others = rng.choice(np.arange(n_main, len(contigs)), size=int(rng.choice([44, 60, 90])) if r in big_at else (1 if tie else int(rng.integers(0, 4))),
                            replace=False)
        for c2 in others.tolist():
            nh = len(mine) if tie else int(rng.choice([1, 1, 2, 3]))
            base = int(rng.integers(10**4, 10**6))
            for q in np.sort(rng.choice(np.arange(100, max(ln, 4000), 7), size=max(nh, 1), replace=False)).tolist():
                per.append((q, c2, base + q, (base + q) // 500 * 500))
        per.sort(key=lambda t: t[0])
        for q, cc, s, g in per:
            name.append(r), pos.append(q), contig.append(cc), start.append(s), ident.append(g)
        rlen.append(ln)
    cols = {k: np.array(v, np.uint32) for k, v in (("name", name), ("pos", pos), ("contig", contig), ("start", start), ("id", ident))}
    # dense key of (contig, ID): what the tagger's loc id is
    _, cols["key"] = np.unique(cols["contig"].astype(np.uint64) << np.uint64(32) | cols["id"], return_inverse=True)
    cols["key"] = cols["key"].astype(np.uint32)
    n_keys = int(cols["key"].max()) + 1
    # read names sort differently from file order (`groupby("rname")` walks them sorted)
    names = [f"r{int(x):06d}" for x in rng.permutation(10**5)[:n_reads]]
    order = np.argsort(np.array(names, dtype=object), kind="stable")
    grank = np.empty(n_reads, np.uint32)
    grank[order] = np.arange(n_reads, dtype=np.uint32)
    in_fai = np.array([0 if j % 9 == 7 else 1 for j in range(len(contigs))], np.uint8)
    return contigs, in_fai, cols, n_keys, grank, (np.array(rlen) >= 10000).astype(np.uint8)


CASE = dict(seed=515, n_reads=240, big_at=(31, 100, 170))


def _single(device=0):
    from gavisunk_amd import engine

    contigs, in_fai, cols, n_keys, grank, long_ = _rows(**CASE)
    rs = engine.RowSet(device)
    rs.append_host(cols["name"], cols["pos"], cols["contig"], cols["start"], cols["id"], cols["key"])
    seg_name, bc, bg, bd, kept = rs.diag_filter(contigs, in_fai)
    rs.keep_filtered()
    bad, meancov, d_bad = engine.RowSet.bad_keys(rs, rs, np.full(n_keys, 3, np.uint8), keep_on_device=True)
    res = rs.pair_stage(grank, long_, d_key_bad=d_bad)
    engine.free_device(d_bad, device)
    rs.close()
    return (seg_name, bc, bg, bd, kept), bad, meancov, res


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gavisunk_amd import dist as gdist
    from gavisunk_amd import engine

    comm = gdist.Comm(0, rccl=False)
    contigs, in_fai, cols, n_keys, grank, long_ = _rows(**CASE)
    lo, hi = gdist.shard_bounds(CASE["n_reads"], world, rank)
    m = (cols["name"] >= lo) & (cols["name"] < hi)
    rs = engine.RowSet(0)
    rs.append_host(cols["name"][m] - np.uint32(lo), cols["pos"][m], cols["contig"][m], cols["start"][m], cols["id"][m], cols["key"][m])
    local, merged = gdist.sharded_second_half(comm, rs, contigs, in_fai, np.full(n_keys, 3, np.uint8), grank[lo:hi], long_[lo:hi], lo)
    rs.close()
    seg_name, bc, bg, bd = local["diag"]
    np.savez(f"{out_path}.{rank}.npz", seg_name=seg_name + np.uint32(lo), bc=bc, bg=bg, bd=bd, kept=local["rows_kept"], bad=local["bad"],
             meancov=np.array(local["meancov"]), cap_in=local["cap_in"], **({f"m_{k}": v for k, v in merged.items()} if merged is not None else {}))
    comm.close()


def _failing_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gavisunk_amd import _lib, engine
    from gavisunk_amd import dist as gdist

    comm = gdist.Comm(0, rccl=False)
    contigs, in_fai, cols, n_keys, grank, long_ = _rows(**CASE)
    lo, hi = gdist.shard_bounds(CASE["n_reads"], world, rank)
    m = (cols["name"] >= lo) & (cols["name"] < hi)
    rs = engine.RowSet(0)
    rs.append_host(cols["name"][m] - np.uint32(lo), cols["pos"][m], cols["contig"][m], cols["start"][m], cols["id"][m], cols["key"][m])
    code = 0
    try:  # rank 1 names too few contigs: its diagonal filter fails (GS_E_ARG) -- and every rank must hear of it
        gdist.sharded_second_half(comm, rs, contigs[:5] if rank == 1 else contigs, in_fai[:5] if rank == 1 else in_fai,
                                  np.full(n_keys, 3, np.uint8), grank[lo:hi], long_[lo:hi], lo)
    except _lib.GsError as e:
        code = e.status
    rs.close()
    np.save(f"{out_path}.{rank}.npy", np.array([code]))
    comm.close()


def test_a_failed_rank_fails_all_ranks(tmp_path):
    """A rank whose filter fails must not leave the others waiting in the next exchange: its status travels with the capacity, and
    every rank raises."""
    from gavisunk_amd import _lib

    out = str(tmp_path / "st")
    mp.spawn(_failing_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert [int(np.load(f"{out}.{r}.npy")[0]) for r in range(3)] == [_lib.GS_E_ARG] * 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_second_half_equals_one_gpu(world, tmp_path):
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (seg_name, bc, bg, bd, kept), bad, meancov, res = _single()
    parts = [np.load(f"{out}.{r}.npz") for r in range(world)]
    # the diagonal filter's verdicts, shard after shard, equal the one-GPU run's -- with a grown Table carried into later shards
    assert [int(p["cap_in"]) for p in parts][0] == 64 and max(int(p["cap_in"]) for p in parts) >= 128
    assert np.array_equal(np.concatenate([p["seg_name"] for p in parts]), seg_name)
    assert np.array_equal(np.concatenate([p["bc"] for p in parts]), bc) and np.array_equal(np.concatenate([p["bg"] for p in parts]), bg)
    ok = bc != 0xFFFFFFFF
    assert np.array_equal(np.concatenate([p["bd"] for p in parts])[ok], bd[ok]) and ok.sum() > 100
    assert sum(int(p["kept"]) for p in parts) == kept
    # bad SUNKs: every rank holds the flags of the whole run
    for p in parts:
        assert np.array_equal(p["bad"], bad) and p["meancov"].tolist() == meancov
    assert 0 < bad.sum() < len(bad)
    # TSV rows and validated intervals on the root
    root = parts[0]
    assert "m_row_id" in root.files and not any("m_row_id" in p.files for p in parts[1:])
    for k in ("row_contig", "row_name", "row_id", "region_contig", "region_start", "region_end", "contigs_in"):
        assert np.array_equal(root[f"m_{k}"], res[k]), k
    assert len(res["row_id"]) > 200 and len(res["region_start"]) > 5


def test_without_the_carried_capacity_the_verdicts_differ():
    """The test above has teeth: a later shard filtered with a fresh 64-slot Table does NOT reproduce the one-GPU verdicts."""
    from gavisunk_amd import engine

    contigs, in_fai, cols, n_keys, grank, long_ = _rows(**CASE)
    (seg_name, bc, bg, bd, kept), *_ = _single()
    lo = 120
    m = cols["name"] >= lo
    outs = []
    for cap_in in (64, 128):
        rs = engine.RowSet(0)
        rs.append_host(cols["name"][m] - np.uint32(lo), cols["pos"][m], cols["contig"][m], cols["start"][m], cols["id"][m], cols["key"][m])
        outs.append(rs.diag_filter(contigs, in_fai, cap_in=cap_in, want_cap=True))
        rs.close()
    want = bc[seg_name >= lo]
    assert np.array_equal(outs[1][1], want) and outs[1][5] == 128
    assert not np.array_equal(outs[0][1], want)


def test_rccl_exchanges_with_one_rank():
    """gs_comm_allreduce_u32 / gs_comm_reduce_u32 / gs_comm_gatherv_host through RCCL with a one-rank communicator (all a one-GPU
    box can start): the sharded second half over the RCCL data plane equals the plain chain."""
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from gavisunk_amd import dist as gdist
        from gavisunk_amd import engine

        comm = gdist.Comm(0, rccl=True)
        contigs, in_fai, cols, n_keys, grank, long_ = _rows(**CASE)
        rs = engine.RowSet(0)
        rs.append_host(cols["name"], cols["pos"], cols["contig"], cols["start"], cols["id"], cols["key"])
        local, merged = gdist.sharded_second_half(comm, rs, contigs, in_fai, np.full(n_keys, 3, np.uint8), grank, long_, 0)
        rs.close()
        got = comm.gatherv(np.arange(10, dtype=np.uint32))
        assert len(got) == 1 and np.array_equal(got[0].view(np.uint32), np.arange(10, dtype=np.uint32))
        # a failed launch of OURS before a collective must fail the call (the slot is looked at before RCCL can write its probe's
        # "invalid argument" into it; only that code is dropped afterwards), and the call after it is clean again
        from gavisunk_amd import _lib
        L = _lib.lib()
        for call in (lambda: comm.gatherv(np.arange(4, dtype=np.uint32)), lambda: comm.exchange_summaries(1, 2)):
            _lib.check(L.gs_selftest_inject_launch_error(0))
            with pytest.raises(_lib.GsError) as e:
                call()
            assert e.value.status == _lib.GS_E_NODEVICE
            call()
        comm.rccl.close()
        comm.rccl = None
    finally:
        dist.destroy_process_group()
    (seg_name, bc, bg, bd, kept), bad, meancov, res = _single()
    assert np.array_equal(local["diag"][1], bc) and np.array_equal(local["bad"], bad) and local["meancov"] == meancov
    for k in ("row_contig", "row_name", "row_id", "region_contig", "region_start", "region_end", "contigs_in"):
        assert np.array_equal(merged[k], res[k]), k


# ---- both halves sharded, from files: every rank tags its block of the reads (carry exchanged), gathers its rows device to device,
# and runs the second half over them; rank 0's merged result must equal one process over the whole file ----
def _files(tmp):
    """A two-haplotype assembly (3 contigs each, shared ancestry, pasted repeats), its SUNK definition through the drop-in
    (`define_sunks`: db + loc files), and 120 reads of haplotype 1."""
    import test_chain_gpu as tc
    from gavisunk_amd import define_sunks, synth

    rng = np.random.default_rng(777)
    haps = tc._assembly(rng, 3, 50_000)
    for h in (0, 1):
        (tmp / f"hap{h + 1}.fa").write_bytes(b"".join(b">%s\n%s\n" % (n.encode(), synth.codes_to_ascii(s)) for n, s in haps[h]))
    assert define_sunks.main(["-k", "20", str(tmp / "hap1.fa"), str(tmp / "hap2.fa"), "--db", str(tmp / "db"), "--loc", str(tmp / "loc")]) == 0
    lens = synth.lognormal_lengths(rng, 120, median=14000, sigma=0.5, lo=3000, hi=45000)
    reads = [synth.codes_to_ascii(r) for r in synth.simulate_reads(rng, [s for _, s in haps[0]], lens)]
    names = [f"read{int(x):05d}" for x in rng.permutation(10**4)[: len(reads)]]
    np.save(tmp / "lens.npy", lens)
    (tmp / "reads.txt").write_bytes(b"\n".join(reads))
    (tmp / "names.txt").write_text("\n".join(names))


def _load(tmp):
    from gavisunk_amd import engine

    db = engine.SunkDB(str(tmp / "db"), str(tmp / "loc"))
    reads = (tmp / "reads.txt").read_bytes().split(b"\n")
    names = (tmp / "names.txt").read_text().split("\n")
    lens = np.load(tmp / "lens.npy")
    order = np.argsort(np.array(names, dtype=object), kind="stable")
    grank = np.empty(len(names), np.uint32)
    grank[order] = np.arange(len(names), dtype=np.uint32)
    table = engine.DeviceTable.from_db(db)
    lr = engine.LocRows.from_db(db)
    kc = lr.key_contig()
    in_fai = np.array([1 if c.startswith("h1") else 0 for c in db.contigs], np.uint8)
    key_hap = np.full(len(kc), 3, np.uint8)  # (this one read set plays both haplotypes' roles in the bad-SUNK step, as in bench.py)
    return db, table, lr, reads, grank, (lens >= 10000).astype(np.uint8), in_fai, key_hap


def _files_worker(rank, world, port, tmp_s, out_path):
    import ctypes as C
    import pathlib

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gavisunk_amd import _lib, engine
    from gavisunk_amd import dist as gdist

    L = _lib.lib()
    comm = gdist.Comm(0, rccl=False)
    db, table, lr, reads, grank, long_, in_fai, key_hap = _load(pathlib.Path(tmp_s))
    lo, hi = gdist.shard_bounds(len(reads), world, rank)
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads[lo:hi]), db.k)
    has, last = C.c_uint32(0), C.c_uint32(0)
    _lib.check(L.gs_tag_phase1(table._h, dev._h, C.byref(has), C.byref(last)))
    summ = comm.exchange_summaries(has.value, last.value)
    _lib.check(L.gs_tag_phase2(table._h, dev._h, gdist.resolve_carry(summ, rank)))
    _lib.check(L.gs_tag_finish(table._h, dev._h, None, 0, None, None, None))
    rs = engine.RowSet(0)
    rs.append_pass(table, dev, 0, lr)  # name ids = read index inside the shard
    local, merged = gdist.sharded_second_half(comm, rs, db.contigs, in_fai, key_hap, grank[lo:hi], long_[lo:hi], lo)
    if merged is not None:
        np.savez(out_path, bad=local["bad"], **merged)
    for x in (rs, dev, lr, table, db):
        x.close()
    comm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_both_halves_sharded_from_files(world, tmp_path):
    from gavisunk_amd import engine

    _files(tmp_path)
    out = str(tmp_path / "merged.npz")
    mp.spawn(_files_worker, args=(world, _free_port(), str(tmp_path), out), nprocs=world, join=True)
    got = np.load(out)
    db, table, lr, reads, grank, long_, in_fai, key_hap = _load(tmp_path)
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads), db.k)
    assert engine.tag(table, dev)[0] == 0
    rs = engine.RowSet(0)
    rs.append_pass(table, dev, 0, lr)
    rs.diag_filter(db.contigs, in_fai)
    rs.keep_filtered()
    bad, _, d_bad = engine.RowSet.bad_keys(rs, rs, key_hap, keep_on_device=True)
    res = rs.pair_stage(grank, long_, d_key_bad=d_bad)
    engine.free_device(d_bad)
    assert np.array_equal(got["bad"], bad) and len(res["row_id"]) > 100 and len(res["region_start"]) >= 3
    for k in ("row_contig", "row_name", "row_id", "region_contig", "region_start", "region_end", "contigs_in"):
        assert np.array_equal(got[k], res[k]), k
    for x in (rs, dev, lr, table, db):
        x.close()
