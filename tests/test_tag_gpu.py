"""Parity of the HIP tagging path (through the C-ABI / the CLI stub) with the reference:
golden outputs of the reference's own binary, and the CPU oracle on seeded random inputs."""
import ctypes as C

import os

import numpy as np
import pytest

from conftest import TAG_CASES
from gavisunk_amd import _lib, engine, kmerpos_annot3, synth
from oracle import oracle, sunkdef

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "compact-hbm", "minimizer", "minimizer-exact", "minimizer-slots", "generic"], autouse=True)
def table_layout(request, monkeypatch):
    """Every test runs six times: with the layout the library picks (the minimiser-addressed table for k >= 16, whatever
    the size -- place records for k <= 21, per-key slots beyond; the generic table below k = 16), with the exact compact copy
    (one bucket load per window: what tables get that the minimiser layout does not take), with the minimiser-addressed layout
    forced (lane = window; "minimizer-exact" runs every tile through the per-lane form that is the fast kernel's fallback;
    "minimizer-slots" = the per-key-slot regions of round 2 also where place records apply), and with the generic
    16-byte-slot table forced.
    (Round 1's small-table layouts -- the LDS join, XCD-partitioned L2 probing -- were retired in round 3: the minimiser kernel
    beats them on their home ground.)"""
    monkeypatch.setenv("GAVISUNK_TABLE", request.param)
    return request.param


def _oracle_hits(k, keys, key_loc_pairs, seqs, prev=oracle.NONE):
    """Oracle run where every key is its own loc row; chrom = 0, group = loc id."""
    t = oracle.Table(k, keys, keys, np.zeros(len(keys), np.uint32), key_loc_pairs.astype(np.int64))
    buf, off = oracle.concat_reads(seqs)
    rc, hits, prev_out, ctr = t.tag(buf, off, prev_row=prev)
    return rc, hits, prev_out, ctr


def _engine_hits(k, keys, loc, seqs, carry=_lib.GS_NONE):
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    batch = engine.ReadBatch.from_ascii(seqs)
    dev = engine.DeviceReads(batch, k)
    st, hits, carry_out, ctr = engine.tag(tab, dev, carry)
    dev.close()
    tab.close()
    return st, hits, carry_out, ctr


def _random_case(seed, k, n_reads=40, glen=30000, median=1500, n_groups_div=50):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, glen, dtype=np.uint8)
    g[5000:6000] = g[1000:2000]  # a repeat: those k-mers are not unique
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], k)
    keys = sunks.code
    loc = (sunks.start // n_groups_div).astype(np.uint32)  # many small location groups
    lens = synth.lognormal_lengths(rng, n_reads, median=median, sigma=0.8, lo=1, hi=glen - 1)
    reads = synth.simulate_reads(rng, [g], lens, sub=0.01, dele=0.01, ins=0.01)
    seqs = [synth.codes_to_ascii(r) for r in reads]
    return keys, loc, seqs


@pytest.mark.parametrize("name", TAG_CASES + ["kat2_missing"])
def test_cli_matches_reference_binary(golden, name, tmp_path):
    c = golden(name)
    for run in c.meta["tag"]:
        out = tmp_path / "out.sunkpos"
        rc = kmerpos_annot3.main([c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]), str(out)])
        assert rc == run["rc"]
        got, exp = out.read_text(), c.text(run["out"])
        if rc == 0:
            assert got == exp, f"{name}: {run['reads']}"
        else:  # the reference died mid-stream: what it had flushed is a prefix of the true hit list
            assert got.startswith(exp)


@pytest.mark.parametrize("name", ["kat1", "amy", "multi_k21"])
def test_cli_in_small_device_batches(golden, name, tmp_path, monkeypatch):
    """The CLI threads `prevLoc` through device batches: tiny batches must not change a byte."""
    monkeypatch.setenv("GAVISUNK_BATCH_BASES", "700")
    c = golden(name)
    run = c.meta["tag"][0]
    out = tmp_path / "out.sunkpos"
    assert kmerpos_annot3.main([c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]), str(out)]) == run["rc"] == 0
    assert out.read_text() == c.text(run["out"])


@pytest.mark.parametrize("name", ["kat1", "kat2", "amy", "multi_k21", "kat2_missing"])
@pytest.mark.parametrize("batch_bases", ["700", "20000"])
def test_cli_host_parser_through_the_streaming_engine(golden, name, batch_bases, tmp_path, monkeypatch, table_layout):
    """GAVISUNK_PARSE=host: batches of the native host reader go through gs_stream_* (upload of the next batch overlapped with
    the pass of the current one, `prevLoc` handed on in device memory) -- the reference binary's bytes, KeyError included."""
    if table_layout not in ("auto", "generic"):
        pytest.skip("one fast and one slow layout are enough here")
    monkeypatch.setenv("GAVISUNK_PARSE", "host")
    monkeypatch.setenv("GAVISUNK_BATCH_BASES", batch_bases)
    c = golden(name)
    for run in c.meta["tag"]:
        out = tmp_path / "out.sunkpos"
        rc = kmerpos_annot3.main([c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]), str(out)])
        assert rc == run["rc"]
        got, exp = out.read_text(), c.text(run["out"])
        assert got == exp if rc == 0 else got.startswith(exp), f"{name}: {run['reads']}"


@pytest.mark.parametrize("name", ["kat1", "multi_k20", "multi_k25", "amy"])
@pytest.mark.parametrize("batch_bases", [None, "900"])
def test_cli_emulates_the_workflows_scatter_chunks(golden, name, batch_bases, tmp_path, monkeypatch):
    """`--emulate-chunks N` on the whole read file = the reference binary run once per scatter chunk, outputs concatenated in
    scatter order (tests/golden/*/expected_chunks<N>.sunkpos, tagONT.smk:2-37,112-131) -- in one run, and also when the file
    spans many device batches (every chunk's `prevLoc` is threaded from batch to batch on its own)."""
    if batch_bases:
        monkeypatch.setenv("GAVISUNK_BATCH_BASES", batch_bases)
    c = golden(name)
    for run in c.meta["tag_chunked"]:
        out = tmp_path / "out.sunkpos"
        rc = kmerpos_annot3.main([c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]), str(out), "--emulate-chunks", str(run["nchunks"])])
        assert rc == 0 and out.read_text() == c.text(run["out"]), f"{name}: {run['nchunks']} chunks"


def test_cli_chunk_emulation_with_one_chunk_is_the_plain_run(golden, tmp_path):
    """one scatter chunk = the plain command, also when it dies of the reference's KeyError half way (the rows before the
    offending window are the output, exit code 1), and with names the device formatter takes from the batch's name blob"""
    for name in ("kat2_missing", "kat1"):
        c = golden(name)
        run = c.meta["tag"][0]
        a, b = tmp_path / f"{name}.plain", tmp_path / f"{name}.chunked"
        rc_a = kmerpos_annot3.main([c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]), str(a)])
        rc_b = kmerpos_annot3.main([c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]), str(b), "--emulate-chunks", "1"])
        assert rc_a == rc_b == run["rc"] and a.read_bytes() == b.read_bytes()
        assert not [f for f in os.listdir(tmp_path) if f.endswith(".chunks.tmp")]  # (the spill file is gone)
    # <out> as a FIFO: the spill file goes to the temporary directory, the chunks' stretches are sent into the pipe
    import threading

    fifo, got = tmp_path / "out.fifo", []
    os.mkfifo(fifo)
    t = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()), daemon=True)
    t.start()
    assert kmerpos_annot3.main([c.path(run["reads"]), c.path(run["db"]), c.path(run["loc"]), str(fifo), "--emulate-chunks", "1"]) == 0
    t.join(timeout=30)
    assert got and got[0] == a.read_bytes()


@pytest.mark.parametrize("k", [11, 16, 18, 20, 21, 23, 25, 27, 31, 32])
@pytest.mark.parametrize("seed", [1, 2])
def test_random_inputs_match_oracle(k, seed):
    keys, loc, seqs = _random_case(seed * 100 + k, k)
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert rc == 0 and st == 0
    assert ctr["windows"] == octr["windows"] and ctr["raw_hits"] == octr["raw_hits"]
    assert len(gh) == len(oh) > 0
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"])
    assert carry == (loc[oprev] if oprev != oracle.NONE else _lib.GS_NONE)


def test_carry_threads_across_batches():
    k = 20
    keys, loc, seqs = _random_case(7, k, n_reads=60)
    _, whole, carry_whole, _ = _engine_hits(k, keys, loc, seqs)
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    carry, parts = _lib.GS_NONE, []
    for lo, hi in ((0, 1), (1, 17), (17, 18), (18, 60)):
        b = engine.ReadBatch.from_ascii(seqs[lo:hi])
        d = engine.DeviceReads(b, k)
        st, h, carry, _ = engine.tag(tab, d, carry)
        h = h.copy()
        h["read"] += lo
        parts.append(h)
        d.close()
    got = np.concatenate(parts)
    assert np.array_equal(got, whole) and carry == carry_whole
    # and a wrong carry-in changes the answer exactly when the first raw hit repeats it
    first_loc = loc[whole["payload"][0]]
    b = engine.ReadBatch.from_ascii(seqs)
    d = engine.DeviceReads(b, k)
    _, h2, _, _ = engine.tag(tab, d, int(first_loc))
    assert len(h2) == len(whole) - 1 or not np.array_equal(h2, whole)
    tab.close()


def test_two_phase_pass_equals_single_pass():
    k = 20
    keys, loc, seqs = _random_case(9, k, n_reads=30)
    _, whole, carry_whole, _ = _engine_hits(k, keys, loc, seqs)
    L = _lib.lib()
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    b = engine.ReadBatch.from_ascii(seqs)
    d = engine.DeviceReads(b, k)
    has, last = C.c_uint32(0), C.c_uint32(0)
    _lib.check(L.gs_tag_phase1(tab._h, d._h, C.byref(has), C.byref(last)))
    assert has.value == 1 and last.value == carry_whole
    _lib.check(L.gs_tag_phase2(tab._h, d._h, _lib.GS_NONE))
    n = C.c_uint64(0)
    _lib.check(L.gs_tag_finish(tab._h, d._h, None, 0, C.byref(n), None, None))
    out = np.zeros(n.value, dtype=engine.HIT_DTYPE)
    _lib.check(L.gs_tag_hits_to_host(tab._h, d._h, out.ctypes.data, n.value, None))
    assert np.array_equal(out, whole)
    tab.close()


def test_many_location_changes_grow_the_candidate_buffer():
    """Every window is a raw hit and consecutive windows alternate between two locations: the
    candidate list is as long as the window list, far beyond the initial budget."""
    k = 20
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, 300_000, dtype=np.uint8)
    codes, first = np.unique(synth.canonical_codes(g, k), return_index=True)
    loc = (first % 2).astype(np.uint32)
    seqs = [synth.codes_to_ascii(g)]
    rc, oh, _, octr = _oracle_hits(k, codes, loc, seqs)
    st, gh, _, ctr = _engine_hits(k, codes, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] == len(g) - k + 1
    assert len(gh) == len(oh) > 250_000
    assert np.array_equal(gh["pos"], oh["pos"]) and np.array_equal(gh["payload"], oh["row"])


def test_empty_and_tiny_batches():
    k = 20
    keys = np.array([0, 5, 77], dtype=np.uint64)  # poly-A is a SUNK here
    loc = np.array([0, 1, 2], dtype=np.uint32)
    st, h, carry, ctr = _engine_hits(k, keys, loc, [])
    assert st == 0 and len(h) == 0 and carry == _lib.GS_NONE
    st, h, carry, ctr = _engine_hits(k, keys, loc, [b"", b"", b"A", b"ACGT" * 10])
    # Q6: empty / short reads yield one A-padded window each -> poly-A hits; repeats are suppressed
    assert h["read"].tolist() == [0] and h["pos"].tolist() == [0] and carry == 0
    assert ctr["windows"] == 3 + (40 - k + 1) and ctr["raw_hits"] == 3


def test_layout_selection(table_layout):
    keys, loc, _ = _random_case(3, 20)
    tab = engine.DeviceTable(20, keys, loc, np.arange(len(keys), dtype=np.uint32))
    want = {"auto": "minimizer-hbm", "compact-hbm": "compact-hbm", "minimizer": "minimizer-hbm", "minimizer-exact": "minimizer-hbm",
            "minimizer-slots": "minimizer-hbm", "generic": "generic"}[table_layout]
    # place records for every k <= 21 table (here ~50 keys per location group: an assembly whose SUNKs sit on point variants, like a
    # diploid one -- such tables are built with hot regions: gs_device.hip, mzp_hot_min); per-key slots only when forced
    assert tab.layout.startswith(want) and ("place records" in tab.layout) == (table_layout in ("auto", "minimizer", "minimizer-exact"))
    tab.close()
    tab = engine.DeviceTable(20, keys, (loc // 1000).astype(np.uint32), np.arange(len(keys), dtype=np.uint32))
    assert tab.layout.startswith(want) and ("place records" in tab.layout) == (table_layout in ("auto", "minimizer", "minimizer-exact"))
    tab.close()
    keys, loc, _ = _random_case(3, 18)
    tab = engine.DeviceTable(18, keys, loc, np.arange(len(keys), dtype=np.uint32))
    assert tab.layout.startswith(want)
    tab.close()
    keys, loc, _ = _random_case(3, 25)
    tab = engine.DeviceTable(25, keys, loc, np.arange(len(keys), dtype=np.uint32))
    # 46-bit keys do not fit 32-bit compact slots without blowing the table up: such tables live in HBM, where the
    # minimiser-addressed layout is the default; forced, the one-bucket-load layout takes 64-bit slots
    if table_layout == "generic":
        assert tab.layout.startswith("generic")
    elif table_layout in ("auto", "minimizer", "minimizer-exact"):  # (k = 22 .. 27: place records under a minimiser of k - 6 bases, w = 7)
        assert tab.layout.startswith("minimizer-hbm") and "place records" in tab.layout and "mod-minimiser of length 19 (w = 7, t = 5)" in tab.layout
    elif table_layout == "minimizer-slots":
        assert tab.layout.startswith("minimizer-hbm") and "50-bit key" in tab.layout and "mod-minimiser of length 21 " in tab.layout
    else:
        assert "compact-hbm" in tab.layout and "x u64" in tab.layout
    tab.close()
    keys, loc, _ = _random_case(3, 11)  # k < 16: only the generic table applies
    tab = engine.DeviceTable(11, keys, loc, np.arange(len(keys), dtype=np.uint32))
    assert tab.layout.startswith("generic")
    tab.close()


def test_spurious_hit_inside_a_run_takes_the_exact_fallback():
    """A read that walks along one location group with single hits of another group sprinkled
    inside it: the partition summaries overlap, so those tiles must go through the generic
    probe; the answer must not change."""
    k = 20
    rng = np.random.default_rng(77)
    g = rng.integers(0, 4, 60_000, dtype=np.uint8)
    codes, first = np.unique(synth.canonical_codes(g, k), return_index=True)
    loc = np.zeros(len(codes), dtype=np.uint32)
    odd = rng.random(len(codes)) < 0.002
    loc[odd] = 1 + (first[odd] % 5).astype(np.uint32)
    seqs = [synth.codes_to_ascii(g), synth.codes_to_ascii(synth.revcomp_codes(g[1000:50_000]))]
    rc, oh, _, octr = _oracle_hits(k, codes, loc, seqs)
    st, gh, _, ctr = _engine_hits(k, codes, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"]
    assert len(gh) == len(oh) > 100
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"])


def test_low_complexity_reads_overflow_one_partition():
    """Homopolymer and short-period repeats put thousands of windows of a tile into ONE key-space partition:
    the partitioned kernel's queue windows, deferred-lookup list and run summaries all overflow their usual sizes."""
    k = 20
    rng = np.random.default_rng(8)
    g = rng.integers(0, 4, 20_000, dtype=np.uint8)
    codes = np.unique(synth.canonical_codes(g, k))
    rep = [b"A" * 9000, b"AC" * 6000, b"ACG" * 3000, b"T" * 5000 + synth.codes_to_ascii(g[:3000]) + b"A" * 4000,
           b"ACGT" * 2500, synth.codes_to_ascii(g), b"GA" * 2048 + b"A" * 64 + b"GA" * 1000]
    extra = np.unique(np.concatenate([synth.canonical_codes(synth.ascii_to_codes(r), k) for r in rep[:5] + rep[6:]]))
    keys = np.unique(np.concatenate([codes, extra]))
    loc = (np.arange(len(keys)) % 3).astype(np.uint32)  # neighbouring windows keep changing location
    rc, oh, _, octr = _oracle_hits(k, keys, loc, rep)
    st, gh, _, ctr = _engine_hits(k, keys, loc, rep)
    assert st == 0 and rc == 0 and ctr["raw_hits"] == octr["raw_hits"] == ctr["windows"]
    assert len(gh) == len(oh) > 1000
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(loc[gh["payload"]], loc[oh["row"]])
    # and with a single location: nearly everything is suppressed, positions deflate by thousands
    loc1 = np.zeros(len(keys), dtype=np.uint32)
    rc, oh, _, _ = _oracle_hits(k, keys, loc1, rep)
    st, gh, _, _ = _engine_hits(k, keys, loc1, rep)
    assert len(gh) == len(oh) == 1 and gh["pos"][0] == oh["pos"][0]


@pytest.mark.parametrize("qb", ["9", "11"])
def test_undersized_compact_table_uses_every_overflow_path(qb, monkeypatch, table_layout):
    """With far too few buckets most keys live in their second-chance bucket or only in the generic side table:
    deferred lookups, their flushes and the generic fallback carry the load, and the answer must not change."""
    if table_layout != "compact-hbm":
        pytest.skip("needs a compact layout")
    monkeypatch.setenv("GAVISUNK_COMPACT_QB", qb)
    k = 20
    keys, loc, seqs = _random_case(55, k, n_reads=60, glen=60000, median=4000)
    loc = (loc % 5).astype(np.uint32)
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    assert f"2^{qb} buckets" in tab.layout and "overflow keys" in tab.layout
    assert int(tab.layout.split("), ")[1].split(" overflow")[0]) > (2000 if qb == "9" else 0)
    tab.close()
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] and len(gh) == len(oh) > 100
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"])


def test_a_few_thousand_tiles():
    k = 20
    keys, loc, seqs = _random_case(77, k, n_reads=500, glen=200000, median=6000, n_groups_div=400)
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["windows"] == octr["windows"] > 2_000_000 and ctr["raw_hits"] == octr["raw_hits"]
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"]) and carry == (loc[oprev] if oprev != oracle.NONE else _lib.GS_NONE)


def test_homopolymer_reads():
    """Long poly-A / poly-T reads: every window is the same k-mer, a SUNK here -- a million raw hits in a row, one region, one
    location; ordinary reads around them."""
    k = 20
    keys, loc, seqs = _random_case(78, k, n_reads=60, glen=60000, median=3000)
    keys = np.concatenate([keys, np.zeros(1, np.uint64)])  # the poly-A k-mer (canonical of poly-T too) is a SUNK here
    loc = np.concatenate([loc, np.array([int(loc.max()) + 1], np.uint32)])
    order = np.argsort(keys, kind="stable")
    keys, loc = keys[order], loc[order]
    seqs = seqs[:20] + [b"A" * 700000, b"T" * 3000 + seqs[20] + b"A" * 50] + seqs[21:40] + [b"T" * 300000] + seqs[40:]
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] > 1_000_000 and len(gh) == len(oh) > 100
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"]) and carry == (loc[oprev] if oprev != oracle.NONE else _lib.GS_NONE)


@pytest.mark.parametrize("k,qb", [(20, "5"), (22, None)])
def test_slots_without_a_spare_bit(k, qb, monkeypatch, table_layout):
    """A single location group and a 31-bit remainder fill a 32-bit slot completely (k = 22 gets there by itself,
    k = 20 with the smallest bucket count): the tag mask is then all 32 bits (found by tools/fuzz_tag.py)."""
    if qb:
        monkeypatch.setenv("GAVISUNK_COMPACT_QB", qb)
    keys, loc, seqs = _random_case(91, k, n_reads=60, glen=8000, median=2000)
    loc = np.zeros_like(loc)
    if table_layout == "compact-hbm":
        tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
        assert "31-bit remainder, 1-bit loc" in tab.layout
        tab.close()
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] > 10000 and len(gh) == len(oh) == 1


@pytest.mark.parametrize("m,load", [("6", "2"), ("12", "8"), ("16", "6"), ("9", "0.5"), ("19", "2"), ("14", "64"), ("15", "16"), ("14", "1")])
def test_minimizer_table_walks_and_overflows(m, load, monkeypatch, table_layout):
    """The minimiser-addressed layouts with the wrong geometry on purpose.  Per-key slots: a short minimiser piles hundreds of
    keys on one 64-byte region (most keys then live in the generic side table: every lookup there is a deferred one); 8 or 6
    keys per 7-slot region make full regions the rule; a sparse table is the easy case.  Place records (m = 14 / 15 at k = 20):
    dozens of places per 7-record region, and a roomy table."""
    place = m in ("14", "15")
    if table_layout not in (("minimizer", "minimizer-exact") if place else ("minimizer-slots",)):
        pytest.skip("the minimiser-addressed layout this geometry belongs to")
    monkeypatch.setenv("GAVISUNK_MZ_M", m)
    monkeypatch.setenv("GAVISUNK_MZ_LOAD", load)
    k = 20
    keys, loc, seqs = _random_case(56, k, n_reads=60, glen=60000, median=4000)
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    assert tab.layout.startswith("minimizer-hbm") and f"mod-minimiser of length {m} " in tab.layout and ("place records" in tab.layout) == place
    n_ovf = int(tab.layout.split(", ")[-1].split(" overflow")[0])
    assert n_ovf > (10000 if m == "6" else 50 if load in ("8", "6", "64") else 20 if load == "16" else -1)
    tab.close()
    seqs = seqs + [b"A" * 5000, b"ACGT" * 700, seqs[0][:19], b""]
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] and ctr["windows"] == octr["windows"] and len(gh) == len(oh) > 100
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"]) and carry == (loc[oprev] if oprev != oracle.NONE else _lib.GS_NONE)


@pytest.mark.parametrize("hot", ["0", "1", "3", "40"])
def test_place_records_hot_regions(hot, monkeypatch, table_layout):
    """Place records with crowded regions on purpose (16 places per 7-record region): a region with at least GAVISUNK_MZ_HOT
    keys that did not fit keeps no record at all -- all its keys live in the generic table and its 60 bytes are a filter of
    them (gs_mzp.h, kMzpHot: what variant-dominated assemblies get by default).  Same rows whatever the threshold."""
    if table_layout not in ("minimizer", "minimizer-exact"):
        pytest.skip("place-record layouts")
    monkeypatch.setenv("GAVISUNK_MZ_M", "15")
    monkeypatch.setenv("GAVISUNK_MZ_LOAD", "16")
    monkeypatch.setenv("GAVISUNK_MZ_HOT", hot)
    k = 20
    keys, loc, seqs = _random_case(57, k, n_reads=60, glen=60000, median=4000)
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    assert "place records" in tab.layout
    assert ("crowded regions kept as filters" in tab.layout) == (hot in ("1", "3")), tab.layout
    tab.close()
    seqs = seqs + [b"A" * 5000, b"ACGT" * 700, seqs[0][:19], b""]
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] and ctr["windows"] == octr["windows"] and len(gh) == len(oh) > 100
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"]) and carry == (loc[oprev] if oprev != oracle.NONE else _lib.GS_NONE)


def test_crowded_minimiser_regions_fall_back_instead_of_failing(monkeypatch, table_layout):
    """The default layout must never abort a table it could have served another way: with 8 keys per 7-slot region far more
    than an eighth of the keys overflow at every minimiser length, so the library walks m upwards and then hands the
    table to the compact / generic layouts (a FORCED minimiser layout reports the misfit instead)."""
    if table_layout not in ("auto", "minimizer"):
        pytest.skip("layout choice of the default path")
    monkeypatch.setenv("GAVISUNK_MZ_LOAD", "16")
    k = 20
    keys, loc, seqs = _random_case(58, k, n_reads=40, glen=2_000_000, median=3000)  # (dozens of places per region at any s)
    if table_layout == "minimizer":
        tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))  # forced + knob: everything may overflow
        assert tab.layout.startswith("minimizer-hbm")
        tab.close()
        return
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    assert not tab.layout.startswith("minimizer-hbm"), tab.layout
    tab.close()
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] and len(gh) == len(oh) > 100
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"]) and np.array_equal(gh["payload"], oh["row"])


@pytest.mark.parametrize("k,m", [(20, 17), (20, 16), (20, 18), (21, 17), (25, 21), (20, 13), (16, 16), (32, 28), (31, 31), (24, 17)])
def test_minimizer_rule_selftest(k, m):
    """The window -> region rule of the minimiser-addressed table has two forms: lane by lane (table build, exact kernel) and
    sliding over the wavefront (k_probe_mz: packed 16-bit minima, halos from the neighbouring steps).  The library compares
    them on random and low-complexity tiles (equal t-mers: the tie rule, both strands): they never differ."""
    mis, bad, hist = C.c_uint64(1), (C.c_uint32 * 4)(), (C.c_uint32 * 64)()
    _lib.check(_lib.lib().gs_selftest_minimizer_rule(0, k, m, 20260 + k * 32 + m, 96, C.byref(mis), bad, hist))
    assert mis.value == 0, (list(bad), list(hist))


@pytest.mark.parametrize("groups_div", [1, 40, 100000])
def test_exact_reads_fill_and_flush_the_hit_list(groups_div, table_layout):
    """Error-free reads: every window is a hit.  The fast minimiser kernel's hit list (512 entries in LDS) is flushed every
    seven steps; with one location group per SUNK every hit is a candidate (2048 per tile: more than the LDS candidate
    buffer, the tile is done over by the exact form), with one group in all none but the first is."""
    k = 20
    rng = np.random.default_rng(77)
    g = rng.integers(0, 4, 50000, dtype=np.uint8)
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], k)
    keys, loc = sunks.code, (sunks.start // groups_div).astype(np.uint32)
    asc = synth.codes_to_ascii(g)
    seqs = [asc[100:9000], asc[20000:20019], asc[30000:49999], synth.codes_to_ascii((3 - g[::-1])[1000:7000])]
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] >= ctr["windows"] - 1  # (the 19-base read is one padded window)
    assert len(gh) == len(oh) and np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
    assert np.array_equal(gh["payload"], oh["row"]) and carry == loc[oprev]
    if table_layout == "minimizer":
        assert (ctr["fallback_tiles"] > 10) == (groups_div == 1)


@pytest.mark.parametrize("k", [20, 25])
def test_more_than_a_million_location_groups(k, table_layout):
    """Loc ids of 21 bits (real assemblies have millions of location groups; the fast minimiser kernel's hit list holds 32-bit
    locs since round 3, before that such a table fell to the exact per-lane kernel) -- and, for k = 25, more than a 64-bit
    slot takes beside the key at all (the minimiser layout does not apply and must not be launched: found by
    tools/fuzz_tag.py seed 9013)."""
    keys, loc, seqs = _random_case(91, k, n_reads=12, glen=1_200_000, median=3000, n_groups_div=1)
    assert int(loc.max()) >= 1 << 20
    seqs = seqs + [seqs[0][:k - 1], b""]
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert st == 0 and ctr["raw_hits"] == octr["raw_hits"] > 1000 and len(gh) == len(oh)
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"]) and np.array_equal(gh["payload"], oh["row"])
    assert carry == (loc[oprev] if oprev != oracle.NONE else _lib.GS_NONE)


def test_table_build_is_deterministic(table_layout, monkeypatch):
    """Tables are built from stable sorts (by minimiser region / by home bucket, then by second-chance bucket): the same keys
    give the same overflow keys -- the ones a crowded region or bucket could not take -- every time, whatever the GPU's
    scheduling did.  (An undersized compact table makes sure there are many of them.)"""
    if table_layout in ("generic", "minimizer-exact"):
        pytest.skip("nothing to overflow")
    if table_layout not in ("minimizer", "auto", "minimizer-slots"):
        monkeypatch.setenv("GAVISUNK_COMPACT_QB", "9")
    elif table_layout != "minimizer-slots":
        monkeypatch.setenv("GAVISUNK_MZ_LOAD", "12")  # (place records: crowded regions, so that keys do overflow)
    keys, loc, seqs = _random_case(57, 20, n_reads=30, glen=300000, median=3000)
    loc = (loc % 7).astype(np.uint32)
    seen, hits = set(), []
    for _ in range(4):
        tab = engine.DeviceTable(20, keys, loc, np.arange(len(keys), dtype=np.uint32))
        seen.add(tab.layout)
        dev = engine.DeviceReads(engine.ReadBatch.from_ascii(seqs), 20)
        st, h, _, _ = engine.tag(tab, dev)
        hits.append(h)
        dev.close()
        tab.close()
    assert len(seen) == 1 and int(next(iter(seen)).split(", ")[-1].split(" overflow")[0]) > 100
    assert all(np.array_equal(h, hits[0]) for h in hits[1:])
    rc, oh, _, _ = _oracle_hits(20, keys, loc, seqs)
    assert len(oh) == len(hits[0]) and np.array_equal(hits[0]["pos"], oh["pos"]) and np.array_equal(hits[0]["payload"], oh["row"])


def test_table_clone_is_the_receiving_side_of_a_broadcast(table_layout):
    """gs_table_clone: what a non-root rank of gs_table_broadcast does (scalars through TableMeta, every array allocated from
    them, the handle initialised without a build), with device-to-device copies -- in every layout; the clone tags like the
    original, also after the original is gone."""
    k = 20
    keys, loc, seqs = _random_case(44, k, n_reads=60, glen=80000, median=2500)
    rc, oh, oprev, octr = _oracle_hits(k, keys, loc, seqs)
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    twin = tab.clone()
    assert twin.layout == tab.layout and twin.nbytes == tab.nbytes
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(seqs), k)
    a = engine.tag(tab, dev)
    tab.close()
    b = engine.tag(twin, dev)
    assert a[0] == b[0] == 0 and np.array_equal(a[1], b[1]) and a[2] == b[2] and b[3]["raw_hits"] == octr["raw_hits"]
    assert len(b[1]) == len(oh) and np.array_equal(b[1]["pos"], oh["pos"]) and np.array_equal(b[1]["payload"], oh["row"])
    dev.close()
    twin.close()


def test_keyerror_after_a_million_rows(table_layout):
    """A db k-mer without loc row met after more than 2^20 emitted rows: the rows before it are handed out -- the reference
    has written them when it dies (kmerpos_annot3.nim:90-93).  (Lists beyond 2^20 rows are not mirrored to the host by the emit
    kernel but fetched in one copy, decided by the pass's total, not by the truncated count.)"""
    if table_layout not in ("auto", "generic"):
        pytest.skip("one fast and one slow layout are enough here")
    k = 20
    rng = np.random.default_rng(12)
    g = rng.integers(0, 4, 1_300_000, dtype=np.uint8)
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], k)
    keys, loc = sunks.code, np.arange(len(sunks), dtype=np.uint32)  # every SUNK its own location: every hit is a row
    bad = int(np.searchsorted(sunks.start, 1_250_000))
    loc[bad] = _lib.GS_MISSING
    seqs = [synth.codes_to_ascii(g), synth.codes_to_ascii(g[:5000])]
    t = oracle.Table(k, keys, keys[loc != _lib.GS_MISSING], np.zeros(len(keys) - 1, np.uint32), loc[loc != _lib.GS_MISSING].astype(np.int64))
    buf, off = oracle.concat_reads(seqs)
    rc, oh, _, _ = t.tag(buf, off, cap=1 << 21)
    st, gh, carry, ctr = _engine_hits(k, keys, loc, seqs)
    assert rc == -2 and st == _lib.GS_E_KEYERROR and len(gh) == len(oh) > (1 << 20)
    assert np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])


def test_table_without_keys():
    st, h, carry, ctr = _engine_hits(20, np.zeros(0, np.uint64), np.zeros(0, np.uint32), [b"ACGT" * 100, b""])
    assert st == 0 and len(h) == 0 and carry == _lib.GS_NONE and ctr["raw_hits"] == 0


def test_device_packing_equals_host_packing():
    k = 20
    keys, loc, seqs = _random_case(31, k, n_reads=50)
    seqs = seqs + [b"", b"acgtn" * 9, b"N" * 70, b"ACGU" * 40]
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    a = engine.tag(tab, engine.DeviceReads(engine.ReadBatch.from_ascii(seqs), k))
    b = engine.tag(tab, engine.DeviceReads.from_ascii(seqs, k))
    assert a[0] == b[0] == 0 and a[1].tobytes() == b[1].tobytes() and a[2] == b[2]
    assert a[3]["windows"] == b[3]["windows"] and a[3]["raw_hits"] == b[3]["raw_hits"]
    tab.close()


def test_deterministic_across_runs():
    k = 20
    keys, loc, seqs = _random_case(21, k, n_reads=50)
    a = _engine_hits(k, keys, loc, seqs)[1]
    b = _engine_hits(k, keys, loc, seqs)[1]
    assert a.tobytes() == b.tobytes()


def test_table_from_db_joins_on_the_device(tmp_path):
    """DeviceTable.from_db (gs_table_create_from_db: the k-mer list joined with the location table by sorts and a merge on the
    device) against the host join of the same files (gs_sunkdb_keys / _key_row / _key_loc) and the oracle: k-mers listed twice in
    <db>, loc rows that repeat a k-mer (the last row wins, kmerpos_annot3.nim:68), loc rows whose k-mer is not in <db>, a db k-mer
    without loc row (KeyError when a read meets it, :90) -- in files big enough for the multi-threaded parsers."""
    from gavisunk_amd import synth

    k = 20
    rng = np.random.default_rng(99)
    g = rng.integers(0, 4, 400_000, dtype=np.uint8)
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], k)
    n = len(sunks.code)
    assert n > 300_000
    bases = ((sunks.code[:, None] >> (2 * (k - 1 - np.arange(k, dtype=np.uint64)))[None, :]) & np.uint64(3)).astype(np.uint8)
    canon = [row.tobytes() for row in np.frombuffer(b"ACGT", np.uint8)[bases]]  # the canonical spelling, as jellyfish / the loc table have it
    dup_db = rng.integers(0, n, 5000)
    missing = b"ACGTTGCATTGACCAGTAGC"
    db_lines = [canon[i] for i in range(n)] + [canon[i] for i in dup_db.tolist()] + [missing]
    order = rng.permutation(len(db_lines))
    (tmp_path / "db").write_bytes(b"".join(db_lines[i] + b"\n" for i in order.tolist()))
    rows = [(i, int(sunks.start[i]), int(sunks.group[i])) for i in range(n)]
    rows += [(int(i), 7_000_000 + j, 9_000_000 + 1000 * (j % 7)) for j, i in enumerate(rng.integers(0, n, 3000).tolist())]  # repeats: the last one wins
    stray = [(b"TTTTTTTTTTGGGGGGGGGG", 5, 5), (b"ACACACACACACACACACAC", 6, 6)]  # in <loc> only: no key
    with open(tmp_path / "loc", "wb") as f:
        for i, s, gr in rows:
            f.write(b"c%d\t%d\t%s\t%d\n" % (s % 3, s, canon[i], gr))
        for km, s, gr in stray:
            f.write(b"c9\t%d\t%s\t%d\n" % (s, km, gr))
    db = engine.SunkDB(str(tmp_path / "db"), str(tmp_path / "loc"))
    t_dev = engine.DeviceTable.from_db(db)            # device join (nothing asked for the host arrays yet)
    t_host = engine.DeviceTable(db.k, db.keys, db.key_loc, db.key_row)  # host join, built from its arrays
    assert len(db.keys) == n + 1 and int((db.key_row == _lib.GS_MISSING).sum()) == 1
    t_dev2 = engine.DeviceTable.from_db(db)           # (after the host join exists: built from it)
    reads = [synth.codes_to_ascii(r) for r in synth.simulate_reads(rng, [g], synth.lognormal_lengths(rng, 40, median=6000, sigma=0.6, lo=50, hi=40000))]
    outs = []
    for t in (t_dev, t_host, t_dev2):
        dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads), k)
        st, hits, carry, ctr = engine.tag(t, dev)
        outs.append((st, hits.copy(), carry, ctr["raw_hits"]))
        dev.close()
    assert outs[0][0] == 0 and len(outs[0][1]) > 500
    for o in outs[1:]:
        assert o[0] == 0 and np.array_equal(o[1], outs[0][1]) and o[2] == outs[0][2] and o[3] == outs[0][3]
    # the winning rows are the LAST ones
    payload_rows = set(outs[0][1]["payload"].tolist())
    assert any(p >= n for p in payload_rows)
    # a read over the k-mer that has no loc row: KeyError from both tables
    for t in (t_dev, t_host):
        dev = engine.DeviceReads(engine.ReadBatch.from_ascii([reads[0][:300] + missing + reads[1][:200]]), k)
        assert engine.tag(t, dev)[0] == _lib.GS_E_KEYERROR
        dev.close()
    for t in (t_dev, t_host, t_dev2):
        t.close()
    db.close()


@pytest.mark.parametrize("k,seed", [(11, 1), (16, 2), (27, 3), (32, 4)])
def test_table_from_db_small_files(tmp_path, k, seed):
    """The same join through the single-threaded line readers (files below 4 MB), other k: random k-mer lists with repeats, loc
    rows with repeats and strays; device join == host join, key for key."""
    rng = np.random.default_rng(seed)
    n = 3000
    codes = np.unique(rng.integers(0, 1 << min(2 * k, 62), n, dtype=np.uint64))
    lut = np.frombuffer(b"ACGT", np.uint8)
    spell = lambda c: lut[((np.uint64(c) >> (2 * (k - 1 - np.arange(k, dtype=np.uint64)))) & np.uint64(3)).astype(np.uint8)].tobytes()
    words = [spell(c) for c in codes.tolist()]
    db_lines = [words[i] for i in rng.integers(0, len(words), len(words) + 500).tolist()] + words[:50]
    (tmp_path / "db").write_bytes(b"".join(w + b"\n" for w in db_lines))
    with open(tmp_path / "loc", "wb") as f:
        for j in range(4000):
            w = words[int(rng.integers(0, len(words)))] if rng.random() < 0.9 else spell(int(rng.integers(0, 1 << min(2 * k, 62))))
            f.write(b"ctg%d\t%d\t%s\t%d\n" % (j % 4, j * 3, w, (j // 40) * 100))
    db = engine.SunkDB(str(tmp_path / "db"), str(tmp_path / "loc"))
    t_dev = engine.DeviceTable.from_db(db)
    t_host = engine.DeviceTable(db.k, db.keys, db.key_loc, db.key_row)
    assert db.k == k and len(db.keys) == len(set(db_lines))
    present = [w for w, r in zip([spell(c) for c in db.keys.tolist()], db.key_row.tolist()) if r != _lib.GS_MISSING]
    reads = [b"".join(present[i] + b"N" for i in rng.integers(0, len(present), 30).tolist()) for _ in range(20)]
    res = []
    for t in (t_dev, t_host):
        dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads), k)
        st, hits, carry, ctr = engine.tag(t, dev)
        res.append((st, hits.copy(), carry, ctr["raw_hits"]))
        dev.close()
        t.close()
    # (at k = 11 a random window can be a db k-mer without loc row: the KeyError is part of what must agree)
    assert res[0][0] == res[1][0] and res[0][0] in (0, _lib.GS_E_KEYERROR)
    assert len(res[0][1]) > 20 and np.array_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:]
    db.close()
