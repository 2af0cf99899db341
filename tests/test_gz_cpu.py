"""The native parallel gzip reader (csrc/gs_inflate.cpp; the `.fq.gz` the workflow feeds the tagger, tagONT.smk:17,36) against
zlib: every kind of deflate block, multi-member files, header fields, block sizes and thread counts give zlib's bytes;
damaged files fail where zlib fails.  Host code only: runs without a GPU."""
import ctypes as C
import gzip
import io
import os
import random
import zlib

import pytest

from gavisunk_amd import _lib, engine


ACGT = bytes(b"ACGT"[i & 3] for i in range(256))


def _inflate(path, threads, want):
    L = _lib.lib()
    z = C.c_void_p()
    _lib.check(L.gs_gz_open(path.encode(), threads, C.byref(z)))
    out = []
    try:
        while True:
            p, n = C.c_void_p(), C.c_uint64(0)
            _lib.check(L.gs_gz_read(z, want, None, 0, C.byref(p), C.byref(n)))
            if n.value:
                out.append(C.string_at(p.value, n.value))
            L.gs_gz_free_buffer(p)
            if not n.value or L.gs_gz_at_end(z):
                break
        d = C.c_uint64(0)
        used = L.gs_gz_pieces(z, C.byref(d))
    finally:
        L.gs_gz_close(z)
    return b"".join(out), used, d.value


def _gz(d, lvl=6, **kw):
    b = io.BytesIO()
    with gzip.GzipFile(fileobj=b, mode="wb", compresslevel=lvl, **kw) as f:
        f.write(d)
    return b.getvalue()


@pytest.fixture(scope="module")
def fastq():
    rnd = random.Random(5)
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, rnd.randbytes(rnd.randint(100, 3000)).translate(ACGT), b"#" * 10) for i in range(2500))


def _cases(fastq):
    rnd = os.urandom(2_000_000)
    c = {
        "empty": _gz(b""), "tiny": _gz(b"hello world\n"), "level6": _gz(fastq * 6, 6), "level1": _gz(fastq * 6, 1), "level9": _gz(fastq * 4, 9),
        "stored": _gz(fastq, 0), "random": _gz(rnd, 6), "zeros": _gz(b"\0" * 30_000_000, 6),
        "multi": _gz(fastq, 1) + _gz(b"", 6) + _gz(fastq * 3, 9) + _gz(b"x", 6) + _gz(rnd, 1) + _gz(fastq * 4, 6),
        "manysmall": b"".join(_gz(fastq[i:i + 60000], 6) for i in range(0, len(fastq), 60000)),
        "named": _gz(fastq * 3, 6, filename="some name.fq", mtime=12345), "padded": _gz(fastq * 3, 6) + b"\0" * 100,
    }
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    c["fixed"] = co.compress(fastq * 2) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for i in range(0, len(fastq), 50000):
        parts += [co.compress(fastq[i:i + 50000]), co.flush(zlib.Z_SYNC_FLUSH)]
    c["syncflush"] = b"".join(parts) + co.flush()
    return c


def test_every_kind_of_stream_inflates_to_zlibs_bytes(tmp_path, fastq):
    path = str(tmp_path / "c.gz")
    parallel = 0
    for name, blob in _cases(fastq).items():
        open(path, "wb").write(blob)
        ref = gzip.decompress(blob[:-100] if name == "padded" else blob)
        for threads, want in ((1, 1 << 62), (3, 1 << 20), (8, 1 << 62)):
            got, used, dropped = _inflate(path, threads, want)
            assert got == ref, (name, threads, want, len(got), len(ref))
        parallel += used > 1
    assert parallel >= 5  # (the dynamic-Huffman files really were cut into several pieces)


def test_damaged_files_fail_like_zlib(tmp_path, fastq):
    blob = _gz(fastq * 4, 6)
    path = str(tmp_path / "c.gz")
    third = len(blob) // 3
    for what, b2 in (("truncated", blob[:len(blob) // 2]), ("crc", blob[:-8] + b"\1\2\3\4" + blob[-4:]), ("isize", blob[:-4] + b"\1\2\3\4"),
                     ("bit flip", blob[:third] + bytes([blob[third] ^ 0x55]) + blob[third + 1:]), ("not gzip", b"hello" * 10)):
        open(path, "wb").write(b2)
        with pytest.raises(Exception):
            gzip.decompress(b2)
        with pytest.raises(_lib.GsError) as e:
            _inflate(path, 4, 1 << 62)
        assert e.value.status == _lib.GS_E_IO, what


def test_text_blocks_of_a_gz_file_are_whole_records(tmp_path, fastq, monkeypatch):
    path = str(tmp_path / "r.fq.gz")
    text = fastq * 5
    open(path, "wb").write(_gz(text, 1))
    monkeypatch.setenv("GAVISUNK_GZ_PIECE_KB", "64")  # (small groups: several reads per file, remainders carried over)
    monkeypatch.setenv("GAVISUNK_GZ_THREADS", "3")
    blocks = [bytes(b) for b in engine.text_blocks(path, 3_000_000)]
    assert b"".join(blocks) == text and len(blocks) >= 2
    assert all(b.count(b"\n") % 4 == 0 and b[:1] == b"@" for b in blocks)
    rnd = random.Random(9)
    fa = b"".join(b">s%d\n%s\n" % (i, rnd.randbytes(rnd.randint(1, 2500)).translate(ACGT)) for i in range(4000))
    open(path, "wb").write(_gz(fa, 6))
    blocks = [bytes(b) for b in engine.text_blocks(path, 2_000_000)]
    assert b"".join(blocks) == fa and len(blocks) >= 2 and all(b[:1] == b">" for b in blocks)
    monkeypatch.setenv("GAVISUNK_GZ", "python")
    assert b"".join(bytes(b) for b in engine.text_blocks(path, 2_000_000)) == fa


def test_fuzz_slice(tmp_path):
    """A seeded slice of tools/fuzz_gz.py (content kinds x levels x strategies x flush patterns x members x threads x piece sizes,
    one file in four damaged): zlib's bytes and newline counts, or an error where zlib refuses.  Seed 71 is the round's find: a
    file cut inside a member whose zero padding decodes as a literal -- the careful loop used to produce output until memory ran
    out; it reports a truncated stream now."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gz

    old = os.environ.get("GAVISUNK_GZ_PIECE_KB")
    try:
        for seed in list(range(60, 85)) + list(range(400, 415)):
            err = fuzz_gz.one_round(seed, str(tmp_path / "f.gz"))
            assert err is None, err
    finally:
        if old is None:
            os.environ.pop("GAVISUNK_GZ_PIECE_KB", None)
        else:
            os.environ["GAVISUNK_GZ_PIECE_KB"] = old


def test_bgzf_shaped_files_are_read_in_parallel_and_in_bounded_calls(tmp_path):
    """bgzip / BGZF -- the usual packaging of ONT .fq.gz -- is a file of thousands of gzip members, each ONE final block of at most
    64 KiB: there is no non-final dynamic block for the finder to find.  Member starts are piece starts too, and a member's end
    is a place to stop a read call: several pieces per call, calls of bounded size (the whole file used to come back from one)."""
    rnd = random.Random(9)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, rnd.randbytes(n := rnd.randint(200, 4000)).translate(ACGT), b"#" * n) for i in range(6000))
    blob = b"".join(_gz(text[i:i + 65280], 6, mtime=0) for i in range(0, len(text), 65280)) + _gz(b"", 6, mtime=0)
    p = tmp_path / "bgzf.fq.gz"
    p.write_bytes(blob)
    L = _lib.lib()
    os.environ["GAVISUNK_GZ_PIECE_KB"] = "256"
    try:
        z = C.c_void_p()
        _lib.check(L.gs_gz_open(str(p).encode(), 4, C.byref(z)))
        want, sizes, out = 2 << 20, [], []
        while True:
            q, n = C.c_void_p(), C.c_uint64(0)
            _lib.check(L.gs_gz_read(z, want, None, 0, C.byref(q), C.byref(n)))
            if n.value:
                out.append(C.string_at(q.value, n.value))
                sizes.append(n.value)
            L.gs_gz_free_buffer(q)
            if not n.value or L.gs_gz_at_end(z):
                break
        d = C.c_uint64(0)
        used = L.gs_gz_pieces(z, C.byref(d))
        L.gs_gz_close(z)
    finally:
        del os.environ["GAVISUNK_GZ_PIECE_KB"]
    assert b"".join(out) == text and len(text) > 20_000_000
    assert len(sizes) >= 4 and max(sizes) < 16 << 20, sizes  # (calls of about `want`, not one call of everything)
    assert used >= 2 * len(sizes)                             # (several pieces per call: the members were found)


def test_runs_and_short_periods(tmp_path, monkeypatch):
    """matches at distance 1, 2 and 3 (the quality line of `seqtk seq -F '#'`, dinucleotide and trinucleotide repeats) are written
    from a register pattern, longer periods copied: the same bytes as zlib, whatever the level and the piece size"""
    rnd = random.Random(7)
    parts = []
    for _ in range(30000):
        k = rnd.randint(0, 4)
        if k == 0:
            parts.append(bytes([rnd.randint(33, 120)]) * rnd.randint(1, 600))
        elif k < 4:
            parts.append(bytes(rnd.choice(b"ACGT") for _ in range(k + 1 if k < 3 else rnd.randint(4, 9))) * rnd.randint(1, 200))
        else:
            parts.append(rnd.randbytes(rnd.randint(1, 50)))
    data = b"".join(parts)
    monkeypatch.setenv("GAVISUNK_GZ_PIECE_KB", "128")
    for lvl in (1, 6, 9):
        p = tmp_path / f"p{lvl}.gz"
        p.write_bytes(_gz(data, lvl))
        for threads in (1, 5):
            got, used, _ = _inflate(str(p), threads, 1 << 20)
            assert got == data and (threads == 1 or used > 1), (lvl, threads)
