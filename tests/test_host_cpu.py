"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/gavisunk.h declares, the native parsers agree with the oracle's, and compute entry
points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, TAG_CASES
from gavisunk_amd import _lib, engine, rlen, synth
from oracle import oracle


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gavisunk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 30
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in _lib.lib().gs_version()


@pytest.mark.parametrize("name", TAG_CASES)
def test_sunkdb_loader_matches_oracle(golden, name):
    c = golden(name)
    run = c.meta["tag"][0]
    db = engine.SunkDB(c.path(run["db"]), c.path(run["loc"]))
    k, codes = oracle.load_db(c.path(run["db"]))
    cn, cid, start, code, group = oracle.load_loc(c.path(run["loc"]))
    assert db.k == k
    assert set(db.keys.tolist()) == set(codes.tolist())
    assert db.contigs == cn
    assert np.array_equal(db.row_contig, cid) and np.array_equal(db.row_start, start)
    assert np.array_equal(db.row_group, group)
    last = {}
    for i, cd in enumerate(code.tolist()):
        last[cd] = i  # later row wins
    for key, row, loc in zip(db.keys.tolist(), db.key_row.tolist(), db.key_loc.tolist()):
        if key in last:
            assert row == last[key]
            assert loc == db.row_loc[row]
        else:
            assert row == _lib.GS_MISSING and loc == _lib.GS_MISSING
    # dense loc ids identify (contig, group) pairs
    pairs = {}
    for l, cc, g in zip(db.row_loc.tolist(), cid.tolist(), group.tolist()):
        assert pairs.setdefault(l, (cc, g)) == (cc, g)
    assert len(set(pairs.values())) == len(pairs)


def _unpack(words, word_off, length, r):
    w = words[int(word_off[r]):int(word_off[r + 1])]
    bases = np.zeros(len(w) * 32, dtype=np.uint8)
    for j in range(32):
        bases[j::32] = (w >> np.uint64(62 - 2 * j)) & np.uint64(3)
    return bases[:length], bases[length:]


@pytest.mark.parametrize("name", ["kat1", "kat2", "short", "multi_k20", "amy"])
def test_fastx_parser_and_packing_match_oracle(golden, name):
    c = golden(name)
    for run in c.meta["tag"]:
        path = c.path(run["reads"])
        names, seqs = oracle.parse_reads(path)
        got_names, got_len, n = [], [], 0
        for batch in engine.read_batches(path, max_bases=50_000, max_reads=7):  # small: many batches
            got_names += batch.names
            got_len += batch.len.tolist()
            for r in range(batch.n_reads):
                bases, pad = _unpack(batch.words, batch.word_off, int(batch.len[r]), r)
                exp = oracle.lib() and np.frombuffer(seqs[n + r], dtype=np.uint8)
                from gavisunk_amd import synth
                assert np.array_equal(bases, synth.ascii_to_codes(exp))
                assert len(pad) >= 32 and not pad.any()  # >= one zero word after each read
            n += batch.n_reads
        assert got_names == names
        assert got_len == [len(s) for s in seqs]


@pytest.mark.parametrize("name", ["kat1", "kat2", "short", "multi_k20", "amy"])
def test_rlen_cli_matches_reference(golden, name, tmp_path):
    c = golden(name)
    for run in c.meta["rlen"]:
        out = tmp_path / "o.rlen"
        assert rlen.main([c.path(run["reads"]), str(out)]) == 0
        assert out.read_text() == c.text(run["out"])


def test_batch_from_ascii_roundtrip():
    rng = np.random.default_rng(0)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTNacgtu", dtype=np.uint8), n).tobytes()) for n in (0, 1, 31, 32, 33, 64, 1000)]
    b = engine.ReadBatch.from_ascii(seqs)
    from gavisunk_amd import synth
    for r, s in enumerate(seqs):
        bases, pad = _unpack(b.words, b.word_off, len(s), r)
        assert np.array_equal(bases, synth.ascii_to_codes(s))
        assert not pad.any()


def test_missing_file_is_an_io_error(tmp_path):
    with pytest.raises(_lib.GsError) as e:
        engine.SunkDB(str(tmp_path / "nope.db"), str(tmp_path / "nope.loc"))
    assert e.value.status == _lib.GS_E_IO
    assert "nope.db" in e.value.msg  # the k-mer list is read first (kmerpos_annot3.nim:20-40): its error wins


def test_loader_errors_name_the_right_file(tmp_path):
    """The two files are parsed by two threads; errors still come out as if the k-mer list had been read first."""
    db, loc = tmp_path / "k.db", tmp_path / "k.loc"
    db.write_text("ACGTACGTACGTACGTACGT\n")
    with pytest.raises(_lib.GsError) as e:
        engine.SunkDB(str(db), str(tmp_path / "nope.loc"))
    assert e.value.status == _lib.GS_E_IO and "nope.loc" in e.value.msg
    loc.write_text("ctg\t12\tACGTACGTACGTACGTACGT\t12\nctg\tnot-a-number\tACGTACGTACGTACGTACGA\t12\n")
    with pytest.raises(_lib.GsError) as e:
        engine.SunkDB(str(db), str(loc))
    assert e.value.status == _lib.GS_E_IO and "k.loc:2" in e.value.msg
    with pytest.raises(_lib.GsError) as e:
        engine.SunkDB(str(tmp_path / "nope.db"), str(loc))
    assert "nope.db" in e.value.msg
    loc.write_text("ctg\t12\tACGTACGTACGTACGTACGT\t12\n")
    d = engine.SunkDB(str(db), str(loc))
    assert d.k == 20 and len(d.keys) == 1 and d.key_row.tolist() == [0] and d.contigs == ["ctg"]
    d.close()


def test_gzip_inputs_whole_file_and_streamed(tmp_path, monkeypatch):
    """.gz inputs are inflated in one go by libdeflate when the system has it (one member or many), streamed through zlib
    otherwise: the parsed batch must not depend on the route."""
    import gzip
    rng = np.random.default_rng(5)
    reads = [rng.integers(0, 4, int(n), dtype=np.uint8) for n in rng.integers(0, 30000, 200)]
    names = [f"r{i} extra words" for i in range(len(reads))]
    plain = tmp_path / "r.fq"
    synth.write_fasta(str(plain), names, reads, fastq=True)
    raw = plain.read_bytes()
    one = tmp_path / "one.fq.gz"
    one.write_bytes(gzip.compress(raw, 1))
    many = tmp_path / "many.fq.gz"  # members cut at arbitrary bytes, like bgzip blocks
    cuts = [0, 1, 70000, 70001, len(raw) // 2, len(raw)]
    many.write_bytes(b"".join(gzip.compress(raw[a:b], 6) for a, b in zip(cuts, cuts[1:])) + gzip.compress(b""))
    empty = tmp_path / "empty.fq.gz"
    empty.write_bytes(gzip.compress(b""))

    def parse(path):
        out = []
        for b in engine.read_batches(str(path), max_bases=1 << 20):
            out.append((b.names, b.len.copy(), b.words.copy(), b.word_off.copy()))
            b.close()
        return out

    want = parse(plain)
    assert sum(len(x[0]) for x in want) == len(reads)
    for route in ("libdeflate", "zlib"):
        if route == "zlib":
            monkeypatch.setenv("GAVISUNK_NO_LIBDEFLATE", "1")
        for path in (one, many):
            got = parse(path)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert g[0] == w[0] and all(np.array_equal(a, b) for a, b in zip(g[1:], w[1:]))
        assert parse(empty) == []


@pytest.mark.skipif(_lib.lib().gs_device_count() > 0, reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_without_gpu():
    keys = np.array([1, 2, 3], dtype=np.uint64)
    with pytest.raises(_lib.GsError) as e:
        engine.DeviceTable(20, keys, np.zeros(3, np.uint32), np.zeros(3, np.uint32))
    assert e.value.status == _lib.GS_E_NODEVICE
    assert "no CPU fallback" in e.value.msg


def test_text_blocks_cut_at_record_boundaries(tmp_path):
    """engine.text_blocks (input of the on-device record parser): blocks are whole records, their concatenation is the file."""
    import gzip

    from gavisunk_amd import engine

    fa = b"".join(b">r%d d\n" % i + b"ACGTACGTAC\n" * (i % 5) for i in range(50))
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * (1 + i % 7), b"@II+" * (1 + i % 7)) for i in range(50))
    for text, first in ((fa, b">"), (fq, b"@"), (fq[:-1], b"@")):
        for gz in (False, True):
            p = tmp_path / ("t.gz" if gz else "t")
            p.write_bytes(gzip.compress(text) if gz else text)
            for max_bytes in (1 << 20, 300, 40, 7):
                blocks = list(engine.text_blocks(str(p), max_bytes))
                assert b"".join(blocks) == text and all(b[:1] == first for b in blocks)
                if first == b"@":
                    assert all(b.count(b"\n") % 4 == 0 for b in blocks[:-1])
                if max_bytes <= 300:
                    assert len(blocks) >= 5
    empty = tmp_path / "e"
    empty.write_bytes(b"")
    assert list(engine.text_blocks(str(empty))) == []


from conftest import DIAG_CASES  # noqa: E402


@pytest.mark.parametrize("name", DIAG_CASES)
def test_native_diag_filter_step2_equals_the_reference_binary(golden, name, tmp_path, capfd):
    """gs_diag_filter_step2_files (the library's own text join, like the reference's compiled program) on the reference
    binaries' own inputs and outputs (tests/golden: `diag_filter_v3` output in, `diag_filter_step2` output expected), to a
    file, through the command-line stub's file-object form, plain and gzipped inputs."""
    import ctypes as C
    import gzip
    import io

    from gavisunk_amd import _lib, diag_filter_step2

    c = golden(name)
    for run in c.meta["diag"]:
        sp, dg, exp = c.path(run["sunkpos"]), c.path(run["out"]), c.text(run["out2"])
        out = tmp_path / "o"
        rows = C.c_uint64(0)
        _lib.check(_lib.lib().gs_diag_filter_step2_files(sp.encode(), dg.encode(), str(out).encode(), C.byref(rows)))
        assert out.read_text() == exp and rows.value == exp.count("\n")
        buf = io.StringIO()
        assert diag_filter_step2.run(sp, dg, out=buf) == 0 and buf.getvalue() == exp
        gz = tmp_path / "s.gz"
        gz.write_bytes(gzip.compress(open(sp, "rb").read()))
        buf = io.StringIO()
        assert diag_filter_step2.run(str(gz), dg, out=buf) == 0 and buf.getvalue() == exp
    with pytest.raises(_lib.GsError):
        diag_filter_step2.run(str(tmp_path / "missing"), dg, out=io.StringIO())


def test_diag_filter_step2_command_line_writes_stdout(golden):
    import subprocess
    import sys

    c = golden("kat3")
    run = c.meta["diag"][0]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "gavisunk_amd.diag_filter_step2", c.path(run["sunkpos"]), c.path(run["out"])],
                       cwd=root, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout == c.text(run["out2"])


def test_loc_table_parsed_by_several_threads_equals_one_thread(tmp_path, monkeypatch):
    """gs_sunkdb_load reads a plain location table of some size with several threads (stretches cut at line ends, contig and
    (contig, group) numbering merged in file order): every array equals the single-thread result, CRLF / empty lines / duplicate
    k-mers (last row wins) included, and a malformed line is reported with the same line number."""
    import random

    from gavisunk_amd import _lib, engine

    rnd = random.Random(3)
    lines = []
    for i in range(160_000):
        c = f"ctg{rnd.randint(0, 30)}"
        s = rnd.randint(0, 10**7)
        lines.append(f"{c}\t{s}\t{''.join(rnd.choice('ACGT') for _ in range(20))}\t{s // 500 * 500}" + ("\r" if i % 7 == 0 else ""))
        if i % 1000 == 0:
            lines.append("")
    lines += lines[100:200]  # duplicate k-mers: the later row wins
    (tmp_path / "loc").write_text("\n".join(lines) + "\n")
    (tmp_path / "db").write_text("\n".join(ln.split("\t")[2] for ln in lines[::3] if ln) + "\n")
    assert (tmp_path / "loc").stat().st_size > (4 << 20)

    def load(threads, loc="loc"):
        if threads:
            monkeypatch.setenv("GAVISUNK_LOC_THREADS", str(threads))
        else:
            monkeypatch.delenv("GAVISUNK_LOC_THREADS", raising=False)
        db = engine.SunkDB(str(tmp_path / "db"), str(tmp_path / loc))
        out = {k: np.array(getattr(db, k)) for k in ("keys", "key_row", "key_loc", "row_contig", "row_start", "row_group", "row_loc")}
        out["contigs"] = list(db.contigs)
        db.close()
        return out

    one = load(1)
    for threads in (0, 3, 7):
        got = load(threads)
        assert got["contigs"] == one["contigs"]
        for k in one:
            if k != "contigs":
                assert np.array_equal(got[k], one[k]), (threads, k)
    assert len(one["row_contig"]) == 160_100 and len(one["contigs"]) == 31
    (tmp_path / "loc_bad").write_text("\n".join(lines[:120_000]) + "\nbad line here\n" + "\n".join(lines[120_000:]) + "\n")
    msgs = []
    for threads in (1, 0):
        with pytest.raises(_lib.GsError) as e:
            load(threads, "loc_bad")
        msgs.append(e.value.msg)
    assert msgs[0] == msgs[1] and ":120001:" in msgs[0]


def test_fuzz_tables_slice(tmp_path, monkeypatch):
    """A seeded slice of tools/fuzz_tables.py: irregular and malformed k-mer lists / location tables through the multi-threaded
    parsers (2, 5, default threads) and the single-threaded readers -- identical arrays, contig order, location ids, identical error
    text -- and the host join against a dictionary."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_tables

    monkeypatch.delenv("GAVISUNK_LOC_THREADS", raising=False)
    try:
        for seed in range(1, 25):
            err = fuzz_tables.one_round(seed, str(tmp_path))
            assert err is None, err
    finally:
        os.environ.pop("GAVISUNK_LOC_THREADS", None)


def _feed_fifo(path, data):
    """a writer of `data` on the far side of a new FIFO (a thread: the reader under test opens the near side)"""
    import threading

    os.mkfifo(path)

    def work():
        try:
            with open(path, "wb") as f:
                f.write(data)
        except BrokenPipeError:  # (the reader closed early: the test's assertion says so)
            pass

    t = threading.Thread(target=work, daemon=True)
    t.start()
    return t


@pytest.mark.timeout(120)
def test_inputs_that_can_be_read_only_once(tmp_path, monkeypatch):
    """The reference opens each of its inputs once (gzopen / open, kmerpos_annot3.nim:20,44,85): a FIFO or a process substitution
    is as good as a file.  Reads (plain and gzip, whatever the name ends with) through the host reader and the text-block reader,
    the k-mer list and the location table above the size where the parallel parsers would open them a second time."""
    import gzip

    rng = np.random.default_rng(5)
    seqs = [synth.codes_to_ascii(rng.integers(0, 4, int(n), dtype=np.uint8)) for n in rng.integers(30, 3000, 400)]
    txt = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"#" * len(s)) for i, s in enumerate(seqs))
    for i, (data, name) in enumerate([(txt, "a.fq"), (txt, "b.fq.gz"), (gzip.compress(txt, 1), "c.fq"), (gzip.compress(txt, 1), "d.fq.gz")]):
        p = str(tmp_path / f"{i}_{name}")
        t = _feed_fifo(p, data)
        got = b"".join(bytes(b) for b in engine.text_blocks(p, 100_000))
        t.join(10)
        assert got == txt, name
        p = str(tmp_path / f"{i}_host_{name}")
        t = _feed_fifo(p, data)
        n, bases = 0, 0
        for b in engine.read_batches(p, max_bases=300_000):
            n += b.n_reads
            bases += int(b.len.sum())
            b.close()
        t.join(10)
        assert (n, bases) == (len(seqs), sum(len(s) for s in seqs)), name
    # the two tables: > 4 MiB each (the parallel parsers' threshold)
    codes = np.unique(rng.integers(0, 1 << 40, 250_000, dtype=np.uint64))
    kmers = [synth.decode_kmer(int(c), 20) for c in codes.tolist()]
    db_txt = "".join(k + "\n" for k in kmers).encode()
    loc_txt = "".join("contig_%d\t%d\t%s\t%d\n" % (i % 3, 7 * i, k, i // 12) for i, k in enumerate(kmers)).encode()
    assert len(db_txt) > (4 << 20) and len(loc_txt) > (4 << 20)
    (tmp_path / "db").write_bytes(db_txt)
    (tmp_path / "loc").write_bytes(loc_txt)
    want = engine.SunkDB(str(tmp_path / "db"), str(tmp_path / "loc"))
    t1, t2 = _feed_fifo(str(tmp_path / "db.fifo"), db_txt), _feed_fifo(str(tmp_path / "loc.fifo"), loc_txt)
    got = engine.SunkDB(str(tmp_path / "db.fifo"), str(tmp_path / "loc.fifo"))
    t1.join(10)
    t2.join(10)
    assert got.k == want.k == 20 and got.contigs == want.contigs
    assert np.array_equal(got.keys, want.keys) and np.array_equal(got.key_row, want.key_row)
    assert np.array_equal(got.row_start, want.row_start) and np.array_equal(got.row_group, want.row_group)
    got.close()
    want.close()


def test_rlen_scanner_equals_the_record_reader(tmp_path, monkeypatch):
    """rlen over text blocks (gs_rowwriter_rlen: a scan of names and line lengths) against the record reader with kseq's rules
    (gs_fastx_*): the same file for every shape; regular shapes are taken by the scanner itself, anything kseq might read differently
    is handed over (GS_E_IRREGULAR) -- never guessed at."""
    import gzip

    rng = np.random.default_rng(9)
    seqs = [synth.codes_to_ascii(rng.integers(0, 4, int(n), dtype=np.uint8)) for n in rng.integers(1, 5000, 300)]
    fq = b"".join(b"@r%d some comment\n%s\n+\n%s\n" % (i, s, bytes(rng.integers(33, 74, len(s), dtype=np.uint8))) for i, s in enumerate(seqs))
    fa1 = b"".join(b">c%d\tdesc\n%s\n" % (i, s) for i, s in enumerate(seqs))
    fa60 = b"".join(b">c%d\n" % i + b"".join(s[a:a + 60] + b"\n" for a in range(0, len(s), 60)) for i, s in enumerate(seqs))
    regular = {
        "fq": fq, "fq_crlf": fq.replace(b"\n", b"\r\n"), "fq_no_last_newline": fq[:-1], "fa_one_line": fa1, "fa_60": fa60,
        "fa_crlf": fa60.replace(b"\n", b"\r\n"), "fa_empty_records": b">a\n>b x\nACGT\n>c\n", "fq_blank_between": fq.replace(b"\n@r7 ", b"\n\n@r7 "),
        "empty": b"", "fq_gz": gzip.compress(fq * 3, 1), "fq_gz_members": gzip.compress(fq, 6) + gzip.compress(fq, 1),
    }
    irregular = {
        "fq_multiline": b"@a\nACGT\nACGT\n+\nIIIIIIII\n@b\nAC\n+\nII\n", "fq_plus_with_name": None,  # (filled below: `+name` lines are regular)
        "fq_empty_seq": b"@r\n\n+\n\n@s\nAC\n+\nII\n", "fq_qual_longer": b"@r\nACGT\n+\nIIIIII\n@s\nAC\n+\nII\n",
        "junk_in_front": b"hello\n" + fq[:2000].rsplit(b"\n@", 1)[0] + b"\n", "fa_then_fq": b">a\nACGT\n@b\nAC\n+\nII\n",
        "fq_truncated": fq[: len(fq) // 2], "header_mid_line": b"xx>a\nACGT\n>b\nAC\n", "fq_qual_starts_with_at": b"@r\nACGT\n+\n@III\n@s\nAC\n+\n@I\n",
    }
    regular["fq_plus_with_name"] = b"".join(b"@r%d\n%s\n+r%d\n%s\n" % (i, s, i, b"#" * len(s)) for i, s in enumerate(seqs[:50]))
    del irregular["fq_plus_with_name"]
    regular["fq_qual_starts_with_at"] = irregular.pop("fq_qual_starts_with_at")  # ('@' is a quality value: four-line records stay regular)
    for name, data in {**regular, **irregular}.items():
        p = tmp_path / (name + (".fq.gz" if "gz" in name else ".txt"))
        p.write_bytes(data)
        monkeypatch.setenv("GAVISUNK_RLEN", "reader")
        assert rlen.main([str(p), str(tmp_path / "want")]) == 0
        monkeypatch.delenv("GAVISUNK_RLEN")
        assert rlen.main([str(p), str(tmp_path / "got")]) == 0
        want = (tmp_path / "want").read_bytes()
        assert (tmp_path / "got").read_bytes() == want, name
        assert rlen._scan(str(p), str(tmp_path / "scan")) == (name in regular), name
        if name in regular:
            assert (tmp_path / "scan").read_bytes() == want
    # blocks smaller than the file: cuts at record ends, every block scanned by itself
    p = tmp_path / "big.fq"
    p.write_bytes(fq * 4)
    with engine.RowWriter(str(tmp_path / "blocks")) as out:
        n = sum(out.rlen(np.frombuffer(b, dtype=np.uint8)) for b in engine.text_blocks(str(p), 50_000))
    assert n == 4 * len(seqs)
    monkeypatch.setenv("GAVISUNK_RLEN", "reader")
    assert rlen.main([str(p), str(tmp_path / "want")]) == 0
    assert (tmp_path / "blocks").read_bytes() == (tmp_path / "want").read_bytes()


def test_rlen_scanner_fuzz_slice(tmp_path, monkeypatch):
    """small FASTQ / FASTA texts with random edits (newlines, record markers, CR, blanks, cuts): whenever the scanner takes a text,
    its file equals the record reader's"""
    rng = np.random.default_rng(123)
    lut = np.frombuffer(b"ACGT", np.uint8)
    repl = [b"\n", b"@", b">", b"+", b"\r", b" ", b"\t", b"\n\n", b"", b"\r\n", b"+\n", b"@x\n"]
    taken = 0
    for it in range(400):
        seqs = [lut[rng.integers(0, 4, int(n), dtype=np.uint8)].tobytes() for n in rng.integers(1, 120, int(rng.integers(1, 8)))]
        if it % 2:
            t = b"".join(b"@r%d c\n%s\n+\n%s\n" % (i, s, bytes(rng.integers(33, 74, len(s), dtype=np.uint8))) for i, s in enumerate(seqs))
        else:
            w = int(rng.choice([0, 30]))
            t = b"".join(b">c%d d\n" % i + (b"".join(s[a:a + w] + b"\n" for a in range(0, len(s), w)) if w else s + b"\n") for i, s in enumerate(seqs))
        t = bytearray(t)
        for _ in range(int(rng.integers(0, 3))):
            at = int(rng.integers(0, len(t)))
            t[at:at + 1] = repl[int(rng.integers(0, len(repl)))]
        if it % 7 == 0:
            t = t[: int(rng.integers(0, len(t) + 1))]
        (tmp_path / "x.txt").write_bytes(bytes(t))
        monkeypatch.setenv("GAVISUNK_RLEN", "reader")
        assert rlen.main([str(tmp_path / "x.txt"), str(tmp_path / "want")]) == 0
        monkeypatch.delenv("GAVISUNK_RLEN")
        if rlen._scan(str(tmp_path / "x.txt"), str(tmp_path / "got")):
            taken += 1
            assert (tmp_path / "got").read_bytes() == (tmp_path / "want").read_bytes(), bytes(t)
    assert 200 < taken < 400
