"""The streaming engine (gs_stream_*, csrc/gs_stream.h): the reference's main loop (workflow/src/kmerpos_annot3.nim:85-96)
as a pipeline of read batches -- upload of batch i + 1 overlapped with the pass of batch i, `prevLoc` (:82-84) handed from
pass to pass in device memory.  Whatever the batching, the rows are those of ONE pass over all the reads = the oracle's."""
import numpy as np
import pytest

from gavisunk_amd import _lib, engine, synth
from oracle import oracle, sunkdef

pytestmark = pytest.mark.gpu

K = 20


def _case(seed, n_reads=300, glen=120000, median=2500, groups_div=40):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, glen, dtype=np.uint8)
    g[50000:53000] = g[10000:13000]
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], K)
    keys, loc = sunks.code, (sunks.start // groups_div).astype(np.uint32)
    lens = synth.lognormal_lengths(rng, n_reads, median=median, sigma=0.9, lo=1, hi=glen - 1)
    seqs = [synth.codes_to_ascii(r) for r in synth.simulate_reads(rng, [g], lens, sub=0.01, dele=0.01, ins=0.01)]
    return keys, loc, seqs


def _oracle(keys, loc, seqs):
    t = oracle.Table(K, keys, keys, np.zeros(len(keys), np.uint32), loc.astype(np.int64))
    buf, off = oracle.concat_reads(seqs)
    rc, oh, prev, ctr = t.tag(buf, off, cap=1 << 22)
    assert rc == 0
    return oh, (loc[prev] if prev != oracle.NONE else _lib.GS_NONE), ctr


def _through_stream(table, seqs, cuts, depth, pinned=False, max_words=1 << 16, max_reads=4096):
    """Batches seqs[cuts[i]:cuts[i+1]] through one stream, submitting as far ahead as the stream allows."""
    st = engine.Stream(table, max_words, max_reads, depth=depth)
    got, keep, raw = [], [], 0
    todo = list(zip(cuts, cuts[1:]))
    flying = []

    def take():
        nonlocal raw
        a = flying.pop(0)
        status, hits, n_reads, ctr = st.next(copy=True)
        assert status == 0 and n_reads == a[1] - a[0]
        hits["read"] += np.uint32(a[0])
        got.append(hits)
        raw += ctr["raw_hits"]

    for a, b in todo:
        batch = engine.ReadBatch.from_ascii(seqs[a:b])
        while True:
            try:
                if pinned:
                    pw = engine.PinnedWords(batch.n_words)
                    pw.words[:] = batch.words
                    st.submit(pw.words, batch.word_off.copy(), batch.len.copy())
                    keep.append(pw)
                else:
                    st.submit_batch(batch)
                break
            except _lib.GsError as e:
                assert e.status == _lib.GS_E_CAPACITY and flying
                take()
        flying.append((a, b))
    while flying:
        take()
    carry = st.carry
    st.close()
    for pw in keep:
        pw.close()
    return (np.concatenate(got) if got else np.zeros(0, engine.HIT_DTYPE)), carry, raw


@pytest.mark.parametrize("depth", [1, 2, 3])
@pytest.mark.parametrize("pinned", [False, True])
def test_stream_equals_one_pass(depth, pinned):
    keys, loc, seqs = _case(5)
    seqs = seqs[:100] + [b"", seqs[100][:K - 1], b"A" * 5000] + seqs[100:]
    oh, ocarry, octr = _oracle(keys, loc, seqs)
    table = engine.DeviceTable(K, keys, loc, np.arange(len(keys), dtype=np.uint32))
    n = len(seqs)
    # tiny, empty and uneven batches; a batch that is one short read; a batch without any read
    cuts = [0, 1, 2, 2, 40, 41, 101, 102, 103, 200, n - 1, n]
    hits, carry, raw = _through_stream(table, seqs, cuts, depth, pinned)
    assert raw == octr["raw_hits"] and carry == ocarry
    assert len(hits) == len(oh) > 500
    assert np.array_equal(hits["read"], oh["read"]) and np.array_equal(hits["pos"], oh["pos"]) and np.array_equal(hits["payload"], oh["row"])
    # and the same as the one-shot path
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(seqs), K)
    st, one, c1, _ = engine.tag(table, dev)
    assert st == 0 and c1 == carry and np.array_equal(one, hits)
    dev.close()
    table.close()


def test_batches_in_flight_and_device_resident_batches():
    """gs_tag_launch_after: three device-resident batches launched back to back (carry handed on in device memory), finished
    out of order; and the same batches through a slot-less stream (gs_stream_submit_reads)."""
    keys, loc, seqs = _case(6, n_reads=240)
    oh, ocarry, octr = _oracle(keys, loc, seqs)
    table = engine.DeviceTable(K, keys, loc, np.arange(len(keys), dtype=np.uint32))
    L = _lib.lib()
    cuts = [0, 80, 81, 240]
    devs = [engine.DeviceReads(engine.ReadBatch.from_ascii(seqs[a:b]), K) for a, b in zip(cuts, cuts[1:])]
    _lib.check(L.gs_tag_launch(table._h, devs[0]._h, _lib.GS_NONE))
    _lib.check(L.gs_tag_launch_after(table._h, devs[1]._h, devs[0]._h))
    _lib.check(L.gs_tag_launch_after(table._h, devs[2]._h, devs[1]._h))
    import ctypes as C

    parts = [None] * 3
    last_carry = None
    for i in (2, 0, 1):
        n, c = C.c_uint64(0), C.c_uint32(0)
        _lib.check(L.gs_tag_finish(table._h, devs[i]._h, None, 0, C.byref(n), C.byref(c), None))
        vp = C.c_void_p()
        _lib.check(L.gs_tag_hits_view(table._h, devs[i]._h, C.byref(vp), None))
        h = engine._view(vp.value, n.value, engine.HIT_DTYPE).copy()
        h["read"] += np.uint32(cuts[i])
        parts[i] = h
        if i == 2:
            last_carry = c.value
    hits = np.concatenate(parts)
    assert last_carry == ocarry and len(hits) == len(oh)
    assert np.array_equal(hits["read"], oh["read"]) and np.array_equal(hits["pos"], oh["pos"]) and np.array_equal(hits["payload"], oh["row"])
    # a stream without slots over the same handles
    st = engine.Stream(table, 0, 0, depth=3)
    for d in devs:
        st.submit_reads(d)
    got = []
    for i in range(3):
        status, h, nr, _ = st.next()
        assert status == 0 and nr == cuts[i + 1] - cuts[i]
        h["read"] += np.uint32(cuts[i])
        got.append(h)
    assert st.carry == ocarry and np.array_equal(np.concatenate(got), hits)
    with pytest.raises(_lib.GsError):
        st.next()
    st.close()
    for d in devs:
        d.close()
    table.close()


def test_stream_keyerror_stops_the_stream(golden, tmp_path):
    """A db k-mer without loc row (kmerpos_annot3.nim:90: KeyError, exit 1): the batch that meets it returns GS_E_KEYERROR with
    the rows before the offending window, and the stream takes no further batch."""
    c = golden("kat2_missing")
    run = c.meta["tag"][0]
    db = engine.SunkDB(c.path(run["db"]), c.path(run["loc"]))
    table = engine.DeviceTable.from_db(db)
    batches = list(engine.read_batches(c.path(run["reads"])))
    st = engine.Stream(table, 1 << 16, 4096, depth=2)
    st.submit_batch(batches[0])
    status, hits, _, _ = st.next()
    assert status == _lib.GS_E_KEYERROR
    # (the reference died mid-stream: what it had flushed is a prefix of the rows before the offending window)
    assert engine.format_sunkpos(hits, batches[0].names, db).startswith(c.text(run["out"])) and len(hits) > 0
    with pytest.raises(_lib.GsError):
        st.submit_batch(batches[0])
    st.close()
    table.close()
    db.close()


def test_stream_batch_limits():
    keys, loc, seqs = _case(7, n_reads=20)
    table = engine.DeviceTable(K, keys, loc, np.arange(len(keys), dtype=np.uint32))
    st = engine.Stream(table, 64, 4, depth=2)
    with pytest.raises(_lib.GsError):
        st.submit_batch(engine.ReadBatch.from_ascii(seqs))  # more reads / words than the slots hold
    st.submit_batch(engine.ReadBatch.from_ascii([b"ACGT" * 10, b""]))
    st.submit_batch(engine.ReadBatch.from_ascii([b"ACGT" * 10]))
    with pytest.raises(_lib.GsError):
        st.submit_batch(engine.ReadBatch.from_ascii([b"ACGT" * 10]))  # two slots, both in flight
    assert st.in_flight == 2
    assert st.next()[0] == 0 and st.next()[0] == 0
    st.close()
    table.close()
