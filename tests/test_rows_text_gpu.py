"""The tagger's output file formatted on the device (csrc/gs_format.hip, gs_rowwriter_rows_device) against the host formatter
(gs_rowwriter_rows) and the oracle's rows: the same bytes of `name \\t pos \\t chrom \\t start \\t group` (kmerpos_annot3.nim:93)."""
import os

import numpy as np
import pytest

from gavisunk_amd import define_sunks, engine, kmerpos_annot3, synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def _case(tmp_path, rng, names):
    fam = rng.integers(0, 4, 400, dtype=np.uint8)
    contigs = []
    for j in range(3):
        s = rng.integers(0, 4, 40_000 + 7000 * j, dtype=np.uint8)
        for p in range(1500, len(s) - 500, 2100):  # a repeat every ~2 kb: the SUNK groups end there, a read crosses several
            s[p:p + 400] = fam
        contigs.append((f"chrUn_{j}_random" if j % 2 else f"c{j}", s))
    (tmp_path / "a.fa").write_bytes(b"".join(b">%s\n%s\n" % (n.encode(), synth.codes_to_ascii(s)) for n, s in contigs))
    assert define_sunks.main(["-k", "20", str(tmp_path / "a.fa"), "--db", str(tmp_path / "db"), "--loc", str(tmp_path / "loc")]) == 0
    lens = synth.lognormal_lengths(rng, len(names), median=6000, sigma=0.6, lo=30, hi=30000)
    reads = synth.simulate_reads(rng, [s for _, s in contigs], lens)
    synth.write_fasta(str(tmp_path / "r.fq"), names, reads, fastq=True)
    return reads


def _run(tmp_path, out, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        assert kmerpos_annot3.main([str(tmp_path / "r.fq"), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / out)]) == 0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return (tmp_path / out).read_bytes()


@pytest.mark.parametrize("kind", ["short", "uuid", "huge"])
def test_device_text_equals_host_text(tmp_path, kind):
    """Short names, ONT-style UUID names, and names so long that a wave's 64 lines do not fit its LDS stage (every lane then writes
    its own line); records parsed on the device (names from the text block) and by the host reader (names from the batch);
    several batches, so that a batch's text starts at any byte offset of the file."""
    rng = np.random.default_rng({"short": 1, "uuid": 2, "huge": 3}[kind])
    n = 180
    if kind == "short":
        names = [f"r{i}" for i in range(n)]
    elif kind == "uuid":
        names = ["%08x-%04x-%04x-%04x-%012x" % tuple(int(x) for x in rng.integers(0, 2**16, 5)) + f"_{i}" for i in range(n)]
    else:
        names = [("N%d_" % i) + "x" * int(rng.integers(100, 700)) for i in range(n)]
    _case(tmp_path, rng, names)
    want = _run(tmp_path, "host.sunkpos", GAVISUNK_ROWS="host")
    assert want.count(b"\n") > 300
    assert _run(tmp_path, "dev.sunkpos") == want
    assert _run(tmp_path, "dev_batches.sunkpos", GAVISUNK_BATCH_BASES="200000") == want
    assert _run(tmp_path, "dev_hostparse.sunkpos", GAVISUNK_PARSE="host") == want
    # and the oracle's rows as text
    db = engine.SunkDB(str(tmp_path / "db"), str(tmp_path / "loc"))
    text = (tmp_path / "r.fq").read_bytes().split(b"\n")
    seqs = text[1::4]
    # (hits of the oracle: `row` = index into the loc rows handed in = here the keys, each with the loc row that won)
    ot = oracle.Table(20, db.keys, db.keys, db.row_contig[db.key_row], db.row_group[db.key_row])
    buf, off = oracle.concat_reads(seqs)
    rc, oh, _, _ = ot.tag(buf, off)
    assert rc == 0
    exp = "".join(f"{names[r]}\t{p}\t{db.contigs[db.row_contig[db.key_row[k]]]}\t{db.row_start[db.key_row[k]]}\t{db.row_group[db.key_row[k]]}\n"
                  for r, p, k in zip(oh["read"].tolist(), oh["pos"].tolist(), oh["row"].tolist()))
    assert exp.encode() == want
    db.close()


def test_device_text_into_a_pipe(tmp_path):
    """<out> may be anything `open(ofile, fmWrite)` takes (kmerpos_annot3.nim:80) -- a FIFO, /dev/stdout into a pipe: an output
    without positions gets the device-formatted pieces one behind the other; same bytes as into a file, several batches."""
    import threading

    rng = np.random.default_rng(4)
    _case(tmp_path, rng, [f"q{i}" for i in range(180)])
    want = _run(tmp_path, "file.sunkpos")
    assert want.count(b"\n") > 300
    fifo = tmp_path / "out.fifo"
    os.mkfifo(fifo)
    got = []

    def reader():
        with open(fifo, "rb") as f:
            got.append(f.read())

    t = threading.Thread(target=reader, daemon=True)
    t.start()
    old = os.environ.get("GAVISUNK_BATCH_BASES")
    os.environ["GAVISUNK_BATCH_BASES"] = "200000"
    try:
        assert kmerpos_annot3.main([str(tmp_path / "r.fq"), str(tmp_path / "db"), str(tmp_path / "loc"), str(fifo)]) == 0
    finally:
        if old is None:
            os.environ.pop("GAVISUNK_BATCH_BASES", None)
        else:
            os.environ["GAVISUNK_BATCH_BASES"] = old
    t.join(timeout=30)
    assert not t.is_alive() and got and got[0] == want


@pytest.mark.parametrize("gz", [False, True])
def test_reads_from_a_pipe(tmp_path, gz):
    """<reads> that can be read only once (a FIFO; the reference's gzopen takes one): the command line reads it through the host
    reader's single open -- the same rows as from the file, plain and gzip, in several batches."""
    import gzip
    import threading

    rng = np.random.default_rng(6)
    _case(tmp_path, rng, [f"p{i}" for i in range(180)])
    want = _run(tmp_path, "file.sunkpos")
    assert want.count(b"\n") > 300
    data = (tmp_path / "r.fq").read_bytes()
    fifo = tmp_path / ("in.fq.gz" if gz else "in.fq")
    os.mkfifo(fifo)

    def feeder():
        try:
            with open(fifo, "wb") as f:
                f.write(gzip.compress(data, 1) if gz else data)
        except BrokenPipeError:
            pass

    t = threading.Thread(target=feeder, daemon=True)
    t.start()
    old = os.environ.get("GAVISUNK_BATCH_BASES")
    os.environ["GAVISUNK_BATCH_BASES"] = "200000"
    try:
        assert kmerpos_annot3.main([str(fifo), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / "pipe.sunkpos")]) == 0
    finally:
        if old is None:
            os.environ.pop("GAVISUNK_BATCH_BASES", None)
        else:
            os.environ["GAVISUNK_BATCH_BASES"] = old
    t.join(timeout=30)
    assert not t.is_alive()
    assert (tmp_path / "pipe.sunkpos").read_bytes() == want
