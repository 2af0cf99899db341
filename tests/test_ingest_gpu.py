"""On-device FASTA / FASTQ record parsing (gs_reads_from_text) against the host parser
(gs_fastx_*: the restatement of the reference's `readfq` loop, kmerpos_annot3.nim:85, pinned by the reference binary's outputs
on tests/golden/*/reads.{fa,fq,fq.gz}): the same text gives the same names, lengths and bases, or the device parser says
"irregular" and the caller falls back -- it never gives different reads."""
import gzip

import numpy as np
import pytest

from conftest import TAG_CASES
from gavisunk_amd import _lib, engine, kmerpos_annot3, synth

pytestmark = pytest.mark.gpu

K = 20


def _host_reads(path):
    names, lens, dev = [], [], []
    for b in engine.read_batches(str(path)):
        names += b.names
        lens += b.len.tolist()
        dev.append(engine.DeviceReads(b, K))
        b.close()
    return names, lens, dev


def _all_windows_table(seqs):
    """Every distinct canonical k-mer of the reads is a key, its own location group: a hit list then spells the bases."""
    from oracle import oracle

    codes = [oracle.canonical_windows(s if len(s) >= K else s + b"A" * (K - len(s)), K) for s in seqs]
    keys = np.unique(np.concatenate(codes)) if codes else np.zeros(0, np.uint64)
    return engine.DeviceTable(K, keys, np.arange(len(keys), dtype=np.uint32) % 1000, np.arange(len(keys), dtype=np.uint32))


def _same_reads(text: bytes, tmp_path, seqs_for_table=None):
    p = tmp_path / "x.txt"
    p.write_bytes(text)
    names, lens, hdev = _host_reads(p)
    d = engine.DeviceReads.from_text(text, K)
    assert d.names == names and d.len.tolist() == lens and d.n_reads == len(names)
    if seqs_for_table is not None and len(names):
        tab = _all_windows_table(seqs_for_table)
        _, hd, cd, ctr_d = engine.tag(tab, d)
        carry, parts, base = _lib.GS_NONE, [], 0
        for h in hdev:
            _, hh, carry, _ = engine.tag(tab, h, carry)
            hh = hh.copy()
            hh["read"] += base
            base += h.n_reads
            parts.append(hh)
        hh = np.concatenate(parts)
        assert len(hd) == len(hh) > 0 and np.array_equal(hd, hh) and cd == carry
        tab.close()
    for h in hdev:
        h.close()
    d.close()
    return names, lens


def _seqs(n=7, seed=1):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        s = bytearray(synth.codes_to_ascii(rng.integers(0, 4, int(rng.integers(1, 900)), dtype=np.uint8)))
        if i % 3 == 1:  # lower case, N, U: `encode` maps them (kmerpos_annot3.nim:24,68)
            s[3:9] = b"acgtnu"
        out.append(bytes(s))
    return out


def test_fasta_forms(tmp_path):
    seqs = _seqs()
    one = b"".join(b">r%d some text\tmore\n%s\n" % (i, s) for i, s in enumerate(seqs))
    names, lens = _same_reads(one, tmp_path, seqs)
    assert names == [f"r{i}" for i in range(len(seqs))] and lens == [len(s) for s in seqs]
    multi = b"".join(b">m%d\n" % i + b"\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + b"\n" for i, s in enumerate(seqs))
    assert _same_reads(multi, tmp_path, seqs)[1] == lens
    assert _same_reads(multi.replace(b"\n", b"\r\n"), tmp_path, seqs)[1] == lens           # CRLF
    assert _same_reads(multi[:-1], tmp_path, seqs)[1] == lens                               # no newline at the end
    assert _same_reads(multi.replace(b"\n>m3", b"\n\n\n>m3") + b"\n\n", tmp_path, seqs)[1] == lens  # empty lines
    empty_rec = b">a\n>b x\nACGT\n>c\n"                                                      # records without bases
    assert _same_reads(empty_rec, tmp_path)[1] == [0, 4, 0]
    assert _same_reads(b">only", tmp_path)[1] == [0]
    assert _same_reads(b">sp  \n AC GT \nNNNN\n", tmp_path, [b" AC GT NNNN"])[1] == [11]      # every byte of a line counts


def test_fastq_forms(tmp_path):
    seqs = _seqs(seed=2)
    rng = np.random.default_rng(5)
    recs = []
    for i, s in enumerate(seqs):
        q = bytes(rng.integers(33, 74, len(s), dtype=np.uint8))
        if i == 2:
            q = b"@" + q[1:]  # a quality line may start with '@' ...
        if i == 4:
            q = b"+" + q[1:]  # ... or '+'
        recs.append(b"@q%d/1 desc\n%s\n+%s\n%s\n" % (i, s, b"q%d" % i if i % 2 else b"", q))
    fq = b"".join(recs)
    names, lens = _same_reads(fq, tmp_path, seqs)
    assert names == [f"q{i}/1" for i in range(len(seqs))] and lens == [len(s) for s in seqs]
    assert _same_reads(fq[:-1], tmp_path, seqs)[1] == lens
    assert _same_reads(fq + b"\n\n", tmp_path, seqs)[1] == lens
    assert _same_reads(fq.replace(b"\n", b"\r\n"), tmp_path, seqs)[1] == lens


@pytest.mark.parametrize("text", [
    b"@r\nACGT\nACGT\n+\nIIII\nIIII\n",          # multi-line FASTQ
    b"@r\n\n+\n\n@s\nAC\n+\nII\n",               # empty sequence
    b"@r\nACGT\n+\nIII\n",                      # short quality
    b"@r\nACGT\n+\nIIII\n\n@s\nAC\n+\nII\n",      # empty line between records
    b">r\nACGT\n@x\nAC\n",                      # '@' line inside FASTA
    b">r\nACGT\n+\nIIII\n",                     # '+' line inside FASTA
    b"\n>r\nACGT\n",                            # does not start with a header
    b"ACGT\n",
    b"@r\nACGT\n+\nIIII\n@s\nAC\n+\n",           # truncated last record
])
def test_irregular_texts_are_refused_not_misread(text, tmp_path, monkeypatch):
    with pytest.raises(_lib.GsError) as e:
        engine.DeviceReads.from_text(text, K)
    assert e.value.status == _lib.GS_E_IRREGULAR
    # ... and the command line stub still gives what the host parser gives for them
    db, loc, reads = tmp_path / "k.db", tmp_path / "k.loc", tmp_path / "r.txt"
    db.write_text("ACGTACGTACGTACGTACGA\n")
    loc.write_text("c\t5\tACGTACGTACGTACGTACGA\t1\n")
    reads.write_bytes(text)
    outs = []
    for mode in ("device", "host"):
        monkeypatch.setenv("GAVISUNK_PARSE", mode)
        assert kmerpos_annot3.main([str(reads), str(db), str(loc), str(tmp_path / f"o_{mode}")]) == 0
        outs.append((tmp_path / f"o_{mode}").read_text())
    assert outs[0] == outs[1]


@pytest.mark.parametrize("name", TAG_CASES)
def test_golden_read_files(golden, name, tmp_path):
    """The reference's own test inputs, every format they come in: device-parsed = host-parsed."""
    c = golden(name)
    for run in c.meta["tag"]:
        path = c.path(run["reads"])
        raw = open(path, "rb").read()
        text = gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw
        p = tmp_path / "t"
        p.write_bytes(text)
        names, lens, hdev = _host_reads(p)
        for h in hdev:
            h.close()
        try:
            d = engine.DeviceReads.from_text(text, K)
        except _lib.GsError as e:
            assert e.status == _lib.GS_E_IRREGULAR
            continue
        assert d.names == names and d.len.tolist() == lens
        d.close()


def test_blocks_of_whole_records(tmp_path):
    seqs = _seqs(n=40, seed=3)
    fa = b"".join(b">r%d\n" % i + b"\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + b"\n" for i, s in enumerate(seqs))
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs))
    for text, gz in ((fa, False), (fq, False), (fq, True), (fa[:-1], True)):
        p = tmp_path / ("x.gz" if gz else "x")
        p.write_bytes(gzip.compress(text) if gz else text)
        for max_bytes in (1 << 30, 5000, 997, 64):
            blocks = list(engine.text_blocks(str(p), max_bytes))
            assert b"".join(blocks) == text and all(b[:1] == text[:1] for b in blocks)
            if max_bytes < 6000:
                assert len(blocks) > 3
            names, lens = [], []
            for b in blocks:
                d = engine.DeviceReads.from_text(b, K)
                names += d.names
                lens += d.len.tolist()
                d.close()
            assert names == [f"r{i}" for i in range(len(seqs))] and lens == [len(s) for s in seqs]


@pytest.mark.parametrize("skew", [1, 7, 15])
def test_text_at_any_alignment(skew, tmp_path, monkeypatch):
    """The line-start kernels read 16 aligned bytes per lane: a text that starts anywhere in such a chunk (and ends anywhere in
    one) gives the same reads -- several 16 KiB tiles, lines that end on tile and chunk boundaries, CRLF, no final newline."""
    monkeypatch.setenv("GAVISUNK_TEST_STAGE_SKEW", str(skew))
    rng = np.random.default_rng(skew)
    seqs = [bytes(synth.codes_to_ascii(rng.integers(0, 4, int(n), dtype=np.uint8))) for n in (16384 - 4 - skew, 1, 15, 16, 17, 4096, 30000, 16384 * 2, 63)]
    fq = b"".join(b"@q%d some words\n" % i + s + b"\n+\n" + b"#" * len(s) + b"\n" for i, s in enumerate(seqs))
    _same_reads(fq, tmp_path, seqs)
    _same_reads(fq[:-1], tmp_path)
    _same_reads(fq.replace(b"\n", b"\r\n"), tmp_path)
    fa = b"".join(b">a%d\n" % i + b"".join(s[j:j + 61] + b"\n" for j in range(0, len(s), 61)) for i, s in enumerate(seqs))
    _same_reads(fa, tmp_path, seqs)
    _same_reads(b">x\n" + b"\n" * 40000 + b"ACGTACGTACGTACGTACGTACGT\n", tmp_path)  # (a newline in every byte of two tiles)


def test_random_texts_never_differ_from_the_host_parser(tmp_path):
    """Random mixtures of the building blocks of both formats, regular or not: refused, or equal."""
    rng = np.random.default_rng(11)
    pieces = [b">h\n", b">h x y\n", b"@h\n", b"@h z\n", b"ACGT\n", b"ACGTACGTAC\n", b"acgtn\n", b"\n", b"+\n", b"+h\n", b"IIII\n",
              b"IIIIIIIIII\n", b"@III\n", b"+III\n", b"\r\n", b"AC\r\n", b"II\r\n", b">\n", b"@\n"]
    refused = equal = 0
    heads, seqls, quals = [b"h", b"h x y", b"h\tz", b""], [b"ACGT", b"ACGTACGTAC", b"acgtn", b"AC"], b"IIIIIIIIII"
    for it in range(600):
        n = int(rng.integers(1, 10))
        if it % 3 == 0:    # any mixture
            lines = [pieces[int(i)] for i in rng.integers(0, len(pieces), n)]
        elif it % 3 == 1:  # FASTA records, multi-line, sometimes empty
            lines = []
            for _ in range(n):
                lines.append(b">" + heads[int(rng.integers(0, 4))] + b"\n")
                lines += [seqls[int(i)] + b"\n" for i in rng.integers(0, 4, int(rng.integers(0, 4)))]
        else:              # four-line FASTQ records
            lines = []
            for _ in range(n):
                sq = seqls[int(rng.integers(0, 4))]
                lines += [b"@" + heads[int(rng.integers(0, 4))] + b"\n", sq + b"\n", b"+\n", quals[:len(sq) + int(rng.integers(0, 2))] + b"\n"]
        if it % 3 and rng.random() < 0.3:  # ... with a stray line somewhere
            lines.insert(int(rng.integers(0, len(lines) + 1)), pieces[int(rng.integers(0, len(pieces)))])
        text = b"".join(lines)
        if rng.random() < 0.3:
            text = text.rstrip(b"\n")
        if not text:
            continue
        try:
            d = engine.DeviceReads.from_text(text, K)
        except _lib.GsError as e:
            assert e.status == _lib.GS_E_IRREGULAR
            refused += 1
            continue
        p = tmp_path / "f"
        p.write_bytes(text)
        names, lens, hdev = _host_reads(p)
        for h in hdev:
            h.close()
        assert d.names == names and d.len.tolist() == lens, text
        d.close()
        equal += 1
    assert refused > 100 and equal > 250
