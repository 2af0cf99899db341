"""GPU SUNK definition (next row N1) against the NumPy restatement of jellyfish -C -U 1 / mrsfast -e 0 /
bedtools merge used for the fixtures (gavisunk_amd/synth.py: define_sunks).  The third-party tools themselves
are not in the build image, so this row is "parity unpinned" beyond that restatement."""
import numpy as np
import pytest

from gavisunk_amd import engine, synth

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
    exp = synth.define_sunks(contigs, k)
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
