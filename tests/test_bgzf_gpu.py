"""DEFLATE on the GPU for BGZF-shaped `.fq.gz` (csrc/gs_bgzf.hip) against zlib: every block type, levels, record cuts, damaged
files; and the command line on such a file against the same reads as plain text."""
import os
import random
import struct
import zlib

import numpy as np
import pytest

from gavisunk_amd import _lib, define_sunks, engine, kmerpos_annot3, synth

pytestmark = pytest.mark.gpu

ACGT = bytes(b"ACGT"[i & 3] for i in range(256))


def bgzf(data: bytes, level: int = 6, block: int = 65280, strategy=zlib.Z_DEFAULT_STRATEGY, eof: bool = True, mem_level: int = 8) -> bytes:
    """bgzip's container: gzip members of at most 64 KiB of text, each with its compressed size in a `BC` extra field."""
    out = []
    for a in list(range(0, len(data), block)) + ([len(data)] if eof else []):
        chunk = data[a:a + block] if a < len(data) else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
        raw = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(raw) + 8
        assert bsize <= 65536
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + raw +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def inflate(path, max_bytes=1 << 26, chunk_kb=512):
    """every text block of the file as bytes (small chunks of compressed bytes, so that a test file makes many blocks)"""
    os.environ["GAVISUNK_BGZF_CHUNK_KB"] = str(chunk_kb)
    try:
        z = engine.BgzfReader(str(path))
        blocks = []
        for st in z.blocks(max_bytes):
            blocks.append(st.host_copy())
            st.close()
        z.close()
    finally:
        del os.environ["GAVISUNK_BGZF_CHUNK_KB"]
    return blocks


@pytest.fixture(scope="module")
def fastq():
    rnd = random.Random(7)
    recs = []
    for i in range(9000):
        n = rnd.randint(150, 9000)
        recs.append(b"@read%d some comment\n%s\n+\n%s\n" % (i, rnd.randbytes(n).translate(ACGT), b"#" * n))
    return b"".join(recs)


@pytest.mark.parametrize("kind", ["level1", "level6", "level9", "stored", "fixed", "huffman_only", "small_blocks"])
def test_inflates_to_zlibs_bytes(tmp_path, fastq, kind):
    kw = {"level1": dict(level=1), "level6": dict(level=6), "level9": dict(level=9), "stored": dict(level=0, block=60000),
          "fixed": dict(level=6, strategy=zlib.Z_FIXED), "huffman_only": dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY, block=40000),
          "small_blocks": dict(level=6, block=777)}[kind]
    data = fastq if kind not in ("small_blocks", "stored") else fastq[: fastq.rfind(b"\n@", 0, 6_000_000) + 1]
    p = tmp_path / "x.fq.gz"
    p.write_bytes(bgzf(data, **kw))
    blocks = inflate(p, 1 << 20)
    assert b"".join(blocks) == data
    assert len(blocks) >= 2 and all(b[:1] == b"@" and b[-1:] == b"\n" for b in blocks)  # whole records, every block
    # every block is made of whole four-line records
    assert all(b.count(b"\n") % 4 == 0 for b in blocks)


def _fuzz_record(rnd: random.Random, i: int) -> bytes:
    """a four-line record whose lines are built to reach every path of the decoder: random bases, tandem repeats of every short
    period (overlapping copies of distance 1 .. 20), long exact repeats (matches of 33 .. 258 bytes), qualities from one repeated
    byte to all 94 printable characters (literal codes longer than the direct table's 8 bits)"""
    n = rnd.choice([1, 2, 7, 40, 300, 2000, 9000, 30000])
    kind = rnd.randrange(5)
    if kind == 0:
        seq = rnd.randbytes(n).translate(ACGT)
    elif kind == 1:
        unit = rnd.randbytes(rnd.randint(1, 20)).translate(ACGT)
        seq = (unit * (n // len(unit) + 1))[:n]
    elif kind == 2:
        piece = rnd.randbytes(rnd.randint(33, 400)).translate(ACGT)
        seq = b"".join(piece if rnd.random() < 0.7 else rnd.randbytes(rnd.randint(1, 50)).translate(ACGT) for _ in range(n // len(piece) + 1))[:n]
    elif kind == 3:
        seq = bytes(rnd.choice(b"ACGTNacgtn") for _ in range(min(n, 3000)))
    else:
        seq = (rnd.randbytes(max(n // 3, 1)).translate(ACGT) * 4)[:n]
    qk = rnd.randrange(4)
    if qk == 0:
        qual = b"#" * len(seq)
    elif qk == 1:
        qual = bytes(33 + (b % 94) for b in rnd.randbytes(len(seq)))
    elif qk == 2:
        qual = bytes(rnd.choice(b"#+5?I") for _ in range(min(len(seq), 4000))).ljust(len(seq), b"I")
    else:
        qual = (bytes(33 + (b % 40) for b in rnd.randbytes(rnd.randint(2, 18))) * (len(seq) // 2 + 1))[:len(seq)]
    name = b"@r%d" % i + (b" " + bytes(33 + (b % 94) for b in rnd.randbytes(rnd.randint(0, 60))) if rnd.random() < 0.5 else b"")
    return name + b"\n" + seq + b"\n+\n" + qual + b"\n"


@pytest.mark.parametrize("seed", range(12))
def test_random_records_levels_strategies_block_sizes(tmp_path, seed):
    """what the encoder can do to the stream -- level, strategy, memory level (1: a new deflate block every 128 symbols, dozens of
    blocks per member), member size -- times what the text can look like: the same bytes as zlib's"""
    rnd = random.Random(1000 + seed)
    data = b"".join(_fuzz_record(rnd, i) for i in range(rnd.randint(150, 500)))
    level = rnd.choice([1, 2, 4, 6, 9])
    strategy = rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED])
    mem_level = rnd.choice([1, 2, 5, 8, 9])
    block = rnd.choice([65280, 65280, 30000, 4096, 300]) if strategy != zlib.Z_HUFFMAN_ONLY else rnd.choice([30000, 4096, 300])
    p = tmp_path / "f.fq.gz"
    p.write_bytes(bgzf(data, level, block, strategy, mem_level=mem_level))
    got = b"".join(inflate(p, rnd.choice([1 << 20, 1 << 22, 1 << 26]), chunk_kb=rnd.choice([128, 512, 4096])))
    assert got == data, (level, strategy, mem_level, block)


def test_fasta_empty_and_tiny(tmp_path):
    rnd = random.Random(3)
    fa = b"".join(b">c%d\n" % i + b"\n".join(rnd.randbytes(60).translate(ACGT) for _ in range(rnd.randint(1, 4000))) + b"\n" for i in range(60))
    (tmp_path / "a.fa.gz").write_bytes(bgzf(fa, 6))
    blocks = inflate(tmp_path / "a.fa.gz", 1 << 21)
    assert b"".join(blocks) == fa and len(blocks) >= 2 and all(b[:1] == b">" for b in blocks)
    (tmp_path / "e.gz").write_bytes(bgzf(b"", 6))
    assert inflate(tmp_path / "e.gz") == []
    (tmp_path / "t.gz").write_bytes(bgzf(b"@r\nACGT\n+\n####\n", 6))
    assert b"".join(inflate(tmp_path / "t.gz")) == b"@r\nACGT\n+\n####\n"


def test_a_read_longer_than_a_block(tmp_path):
    rnd = random.Random(4)
    n = 9_000_000
    data = b"@big\n" + rnd.randbytes(n).translate(ACGT) + b"\n+\n" + b"#" * n + b"\n@small\nACGT\n+\n####\n"
    (tmp_path / "b.fq.gz").write_bytes(bgzf(data, 1))
    assert b"".join(inflate(tmp_path / "b.fq.gz", 1 << 20)) == data


def test_damaged_files_fail_like_zlib(tmp_path, fastq):
    good = bgzf(fastq[:8_000_000], 6)
    cases = {}
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x10
    cases["flip"] = bytes(bad)
    cases["trunc"] = good[: len(good) * 2 // 3]
    crc = bytearray(good)
    crc[70000] ^= 1  # (somewhere inside a member's deflate data or trailer: either way the member no longer checks out)
    cases["flip2"] = bytes(crc)
    for name, blob in cases.items():
        p = tmp_path / f"{name}.gz"
        p.write_bytes(blob)
        with pytest.raises(_lib.GsError) as e:
            inflate(p)
        assert e.value.status == _lib.GS_E_IO, name
    # a plain gzip file is not BGZF: refused at open, for the host reader to take
    import gzip
    (tmp_path / "plain.gz").write_bytes(gzip.compress(fastq[:100000]))
    with pytest.raises(_lib.GsError) as e:
        engine.BgzfReader(str(tmp_path / "plain.gz"))
    assert e.value.status == _lib.GS_E_IRREGULAR


def test_command_line_on_a_bgzf_file(tmp_path):
    rng = np.random.default_rng(12)
    fam = rng.integers(0, 4, 400, dtype=np.uint8)
    contigs = []
    for j in range(3):
        s = rng.integers(0, 4, 50_000, dtype=np.uint8)
        for p in range(1500, len(s) - 500, 2100):
            s[p:p + 400] = fam
        contigs.append((f"c{j}", s))
    (tmp_path / "a.fa").write_bytes(b"".join(b">%s\n%s\n" % (n.encode(), synth.codes_to_ascii(s)) for n, s in contigs))
    assert define_sunks.main(["-k", "20", str(tmp_path / "a.fa"), "--db", str(tmp_path / "db"), "--loc", str(tmp_path / "loc")]) == 0
    lens = synth.lognormal_lengths(rng, 400, median=7000, sigma=0.6, lo=30, hi=40000)
    reads = synth.simulate_reads(rng, [s for _, s in contigs], lens)
    names = [f"m{i}/{int(rng.integers(0, 10**6))}" for i in range(len(reads))]
    synth.write_fasta(str(tmp_path / "r.fq"), names, reads, fastq=True)
    text = (tmp_path / "r.fq").read_bytes()
    (tmp_path / "r.fq.gz").write_bytes(bgzf(text, 6))
    assert kmerpos_annot3.main([str(tmp_path / "r.fq"), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / "plain.sunkpos")]) == 0
    assert kmerpos_annot3.main([str(tmp_path / "r.fq.gz"), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / "gz.sunkpos")]) == 0
    want = (tmp_path / "plain.sunkpos").read_bytes()
    assert want.count(b"\n") > 500 and (tmp_path / "gz.sunkpos").read_bytes() == want
    assert "read_and_inflate_on_device" in kmerpos_annot3.last_timing
    os.environ["GAVISUNK_BATCH_BASES"] = "300000"  # several text blocks: the record cut and the carried bytes
    os.environ["GAVISUNK_BGZF_CHUNK_KB"] = "128"
    try:
        assert kmerpos_annot3.main([str(tmp_path / "r.fq.gz"), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / "gz2.sunkpos")]) == 0
    finally:
        del os.environ["GAVISUNK_BATCH_BASES"], os.environ["GAVISUNK_BGZF_CHUNK_KB"]
    assert (tmp_path / "gz2.sunkpos").read_bytes() == want


def test_a_file_that_goes_on_as_plain_gzip(tmp_path):
    """bgzip members followed by plain gzip members are one valid gzip file: the reader refuses it where the container ends (it
    never misreads), and the command line starts over with the host reader -- the rows of the plain text."""
    import gzip

    rng = np.random.default_rng(5)
    fam = rng.integers(0, 4, 400, dtype=np.uint8)
    contigs = []
    for j in range(3):
        c = rng.integers(0, 4, 50_000, dtype=np.uint8)
        for at in range(1500, len(c) - 500, 2100):
            c[at:at + 400] = fam
        contigs.append(c)
    (tmp_path / "a.fa").write_bytes(b"".join(b">c%d\n%s\n" % (j, synth.codes_to_ascii(c)) for j, c in enumerate(contigs)))
    assert define_sunks.main(["-k", "20", str(tmp_path / "a.fa"), "--db", str(tmp_path / "db"), "--loc", str(tmp_path / "loc")]) == 0
    reads = synth.simulate_reads(rng, contigs, synth.lognormal_lengths(rng, 300, median=7000, sigma=0.6, lo=30, hi=40000))
    synth.write_fasta(str(tmp_path / "r.fq"), [f"q{i}" for i in range(len(reads))], reads, fastq=True)
    text = (tmp_path / "r.fq").read_bytes()
    cut = text.rfind(b"\n@q", 0, len(text) // 2) + 1
    mixed = bgzf(text[:cut], 6, eof=False) + gzip.compress(text[cut:], 6)
    assert gzip.decompress(mixed) == text
    (tmp_path / "m.fq.gz").write_bytes(mixed)
    with pytest.raises(_lib.GsError) as e:
        inflate(tmp_path / "m.fq.gz", 1 << 18, chunk_kb=128)
    assert e.value.status == _lib.GS_E_IRREGULAR
    os.environ["GAVISUNK_BATCH_BASES"] = "100000"  # (several blocks: rows are written before the container ends)
    os.environ["GAVISUNK_BGZF_CHUNK_KB"] = "128"
    try:
        assert kmerpos_annot3.main([str(tmp_path / "m.fq.gz"), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / "m.sunkpos")]) == 0
    finally:
        del os.environ["GAVISUNK_BATCH_BASES"], os.environ["GAVISUNK_BGZF_CHUNK_KB"]
    assert kmerpos_annot3.main([str(tmp_path / "r.fq"), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / "p.sunkpos")]) == 0
    want = (tmp_path / "p.sunkpos").read_bytes()
    assert want.count(b"\n") > 100 and (tmp_path / "m.sunkpos").read_bytes() == want
    assert "read_or_inflate_text" in kmerpos_annot3.last_timing
