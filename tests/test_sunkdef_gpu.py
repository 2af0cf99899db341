"""GPU SUNK definition (next row N1: gs_sunks_define, the `define_sunks` command) against TWO independent checkers: the oracle's
NumPy restatement of jellyfish -C -U 1 / mrsfast -e 0 / bedtools merge (oracle/sunkdef.py, line-cited to
workflow/rules/defineSUNKs.smk) and a brute-force dictionary counter written here, in plain Python, from the rules' semantics
alone.  The third-party tools themselves are not in the build image and the reference holds no fixture for this stage, so the
row stays "parity unpinned" by rule -- but the GPU path is no longer checked against product code or against one restatement."""
import numpy as np
import pytest

from gavisunk_amd import engine, synth
from oracle import sunkdef

pytestmark = pytest.mark.gpu


def _contigs(seed):
    rng = np.random.default_rng(seed)
    fam = rng.integers(0, 4, 700, dtype=np.uint8)
    out = []
    for i, n in enumerate((30000, 1, 19, 20, 21, 12000, 0, 45000)):
        s = rng.integers(0, 4, n, dtype=np.uint8)
        for _ in range(n // 6000):
            p = int(rng.integers(0, max(n - 700, 1)))
            s[p:p + 700] = fam[: len(s[p:p + 700])]
        a = bytearray(synth.codes_to_ascii(s))
        for p in rng.integers(0, max(n, 1), n // 3000):
            a[int(p)] = ord("N")
        if i % 2:
            a = bytearray(bytes(a).lower())
        out.append((f"c{i}", bytes(a)))
    out.append(("rc_of_c0", synth.codes_to_ascii(synth.revcomp_codes(synth.ascii_to_codes(out[0][1][:5000].upper().replace(b"N", b"A"))))))
    return out


@pytest.mark.parametrize("k", [16, 20, 21, 31, 32])
def test_matches_numpy_restatement(k):
    contigs = _contigs(k)
    exp = sunkdef.define_sunks(contigs, k)
    got = engine.define_sunks(contigs, k)
    assert len(got) == len(exp) > 1000
    assert np.array_equal(got.contig_id, exp.contig_id) and np.array_equal(got.start, exp.start)
    assert np.array_equal(got.code, exp.code) and np.array_equal(got.group, exp.group)


def test_defined_table_tags_like_the_file_route(tmp_path):
    from gavisunk_amd import define_sunks as cli, kmerpos_annot3

    contigs = _contigs(3)
    fa = tmp_path / "asm.fa"
    fa.write_bytes(b"".join(b">" + n.encode() + b"\n" + s + b"\n" for n, s in contigs))
    assert cli.main(["-k", "20", str(fa), "--db", str(tmp_path / "db"), "--loc", str(tmp_path / "loc")]) == 0
    rng = np.random.default_rng(1)
    codes = [synth.ascii_to_codes(s.upper()) for _, s in contigs if len(s) > 5000]
    reads = synth.simulate_reads(rng, codes, synth.lognormal_lengths(rng, 30, median=3000, lo=100, hi=9000))
    synth.write_fasta(str(tmp_path / "r.fa"), [f"r{i}" for i in range(len(reads))], reads)
    assert kmerpos_annot3.main([str(tmp_path / "r.fa"), str(tmp_path / "db"), str(tmp_path / "loc"), str(tmp_path / "o")]) == 0
    from oracle import oracle
    rc, txt = oracle.tag_files(str(tmp_path / "r.fa"), str(tmp_path / "db"), str(tmp_path / "loc"))
    assert rc == 0 and (tmp_path / "o").read_text() == txt and txt.count("\n") > 20


# ---- a second, independent checker: a dictionary over every window, nothing shared with oracle/sunkdef.py ---------------------
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _brute_force(contigs, k):
    """[(contig, start, canonical k-mer, group)] in (contig order, start) order, straight from the rules' semantics: a k-mer is
    counted with its reverse complement (jellyfish -C), windows with a non-ACGT byte are skipped, SUNK = count 1 over all
    contigs (-U 1), consecutive SUNK starts on a contig that are at most k apart share a group = the first start of the run
    (bedtools merge joins overlapping and book-ended [start, start + k) intervals)."""
    count, where = {}, {}
    for name, seq in contigs:
        s = seq.decode().upper()
        for i in range(len(s) - k + 1):
            w = s[i:i + k]
            if any(ch not in _COMP for ch in w):
                continue
            rc = "".join(_COMP[ch] for ch in reversed(w))
            c = min(w, rc)
            count[c] = count.get(c, 0) + 1
            where[c] = (name, i)
    order = {name: j for j, (name, _) in enumerate(contigs)}
    rows = sorted((order[nm], i, c) for c, (nm, i) in where.items() if count[c] == 1)
    out, prev, grp = [], None, None
    for ci, i, c in rows:
        if prev is None or prev[0] != ci or i - prev[1] > k:
            grp = i
        out.append((contigs[ci][0], i, c, grp))
        prev = (ci, i)
    return out


_TINY = [
    # overlapping and book-ended SUNK intervals, a gap of exactly k and of k + 1 bases of repeat between unique stretches
    ("unique_then_repeat", lambda r: r(60) + b"ACGTTGCA" * 6 + r(45) + b"ACGTTGCA" * 6 + r(30)),
    ("n_inside", lambda r: r(40) + b"N" + r(39) + b"nn" + r(25) + b"N"),
    ("lower_case", lambda r: r(70).lower()),
    ("shorter_than_k", lambda r: r(7)),
    ("exactly_k", lambda r: r(20)[:20]),
    ("empty", lambda r: b""),
    ("homopolymers", lambda r: b"A" * 50 + r(30) + b"T" * 50),
]


@pytest.mark.parametrize("k", [8, 20])
def test_matches_a_brute_force_dictionary_counter(k, tmp_path):
    import random

    from gavisunk_amd import define_sunks as cli

    rnd = random.Random(5)
    mk = lambda n: bytes(rnd.choice(b"ACGT") for _ in range(n))  # noqa: E731
    contigs = [(nm, f(mk)) for nm, f in _TINY]
    # a contig that repeats a stretch of another one as its REVERSE COMPLEMENT: those k-mers count twice under -C
    src = contigs[0][1][:50].decode().upper()
    contigs.append(("revcomp_duplicate", "".join(_COMP[c] for c in reversed(src)).encode() + mk(30)))
    want = _brute_force(contigs, k)
    assert len(want) > 100
    got = engine.define_sunks(contigs, k)
    names = [n for n, _ in contigs]
    rows = [(names[c], int(s), synth.decode_kmer(int(cd), k), int(g)) for c, s, cd, g in zip(got.contig_id, got.start, got.code, got.group)]
    assert rows == want
    # the command-line stub's files, byte for byte: db = the sorted k-mer list, loc = the rows above as text
    fa = tmp_path / "asm.fa"
    fa.write_bytes(b"".join(b">" + n.encode() + b"\n" + s + b"\n" for n, s in contigs if len(s)) + b">empty\n")
    assert cli.main(["-k", str(k), str(fa), "--db", str(tmp_path / "db"), "--loc", str(tmp_path / "loc")]) == 0
    assert sorted((tmp_path / "db").read_text().split()) == sorted(r[2] for r in want)
    assert (tmp_path / "loc").read_text() == "".join(f"{n}\t{s}\t{c}\t{g}\n" for n, s, c, g in want)
