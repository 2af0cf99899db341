"""Bad-SUNK caller (next row N2): GPU histogram behind the badsunks stub vs the oracle restatement."""
import numpy as np
import pytest

from gavisunk_amd import badsunks
from oracle import oracle

pytestmark = pytest.mark.gpu


def _synthetic(seed, own, other):
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(600):
        for _ in range(int(rng.integers(1, 30))):
            c = own[int(rng.integers(0, len(own)))] if rng.random() < 0.8 else other[int(rng.integers(0, len(other)))]
            g = int(rng.zipf(1.4)) % 400 * 1000
            rows.append(f"r{r}\t{int(rng.integers(0, 50000))}\t{c}\t{g + 5}\t{g}\n")
    return "".join(rows)


def test_matches_oracle(tmp_path, golden):
    fai1 = "h1a\t10\nh1b\t10\n"
    fai2 = "h2a\t10\n"
    sp1 = _synthetic(1, ["h1a", "h1b"], ["h2a"])
    sp2 = _synthetic(2, ["h2a"], ["h1a", "h1b"])
    p = {n: tmp_path / n for n in ("f1", "f2", "s1", "s2", "out")}
    p["f1"].write_text(fai1), p["f2"].write_text(fai2), p["s1"].write_text(sp1), p["s2"].write_text(sp2)
    assert badsunks.main([str(p["f1"]), str(p["f2"]), str(p["s1"]), str(p["s2"]), str(p["out"])]) == 0
    exp = oracle.badsunks(fai1, fai2, sp1, sp2)
    assert p["out"].read_text() == "".join(e + "\n" for e in sorted(exp)) and len(exp) > 20  # the file, sorted (the script's set order is arbitrary)
    # and on a reference-made .sunkpos (AMY locus): hap1 reads against both haplotype roles
    c = golden("amy")
    sp = c.text("expected.sunkpos")
    fai = c.text("fai")
    p["s1"].write_text(sp), p["f1"].write_text(fai), p["f2"].write_text("AMY_CHM1\t467195\t10\t60\t61\n"), p["s2"].write_text(sp)
    assert badsunks.main([str(p["f1"]), str(p["f2"]), str(p["s1"]), str(p["s2"]), str(p["out"])]) == 0
    assert p["out"].read_text() == "".join(e + "\n" for e in sorted(oracle.badsunks(fai, "AMY_CHM1\t467195\t10\t60\t61\n", sp, sp)))


def test_contig_names_with_colons(tmp_path):
    """`ID2.str.split(":", 1)` (badsunks_AR.py:29): the contig looked up in the fai is the text before the FIRST colon of
    `chrom:ID`.  A contig called chr1:1-5000 is therefore looked up as `chr1`: with `chr1` in the fai it counts as this
    haplotype's, with only `chr1:1-5000` in the fai it does not."""
    for fai1, fai2 in (("chr1\t10\n", "chr2\t10\n"), ("chr1:1-5000\t10\n", "chr2:7-9\t10\n")):
        sp1 = _synthetic(5, ["chr1:1-5000"], ["chr2:7-9"])
        sp2 = _synthetic(6, ["chr2:7-9"], ["chr1:1-5000"])
        p = {n: tmp_path / n for n in ("f1", "f2", "s1", "s2", "out")}
        p["f1"].write_text(fai1), p["f2"].write_text(fai2), p["s1"].write_text(sp1), p["s2"].write_text(sp2)
        if fai1.startswith("chr1:"):  # no SUNK is on "this haplotype": the script dies on mode().values[0], so does the stub
            with pytest.raises(SystemExit):
                badsunks.main([str(p["f1"]), str(p["f2"]), str(p["s1"]), str(p["s2"]), str(p["out"])])
            continue
        assert badsunks.main([str(p["f1"]), str(p["f2"]), str(p["s1"]), str(p["s2"]), str(p["out"])]) == 0
        exp = oracle.badsunks(fai1, fai2, sp1, sp2)
        assert p["out"].read_text() == "".join(e + "\n" for e in sorted(exp)) and len(exp) > 5


@pytest.mark.parametrize("name", ["synth", "coverage", "amy", "colon_first_part_in_fai", "colon_whole_name_in_fai"])
def test_matches_the_reference_script(name, tmp_path, capsys):
    """tests/golden/badsunks/*: what the reference's own badsunks_AR.py wrote for these inputs (make_golden_badsunks.py runs it in
    place through the pandas-1.3 naming adapter): the same set, the same two "mean coverage" values, the same failure when no
    SUNK counts as the haplotype's own."""
    import json
    import os

    from conftest import ROOT

    d = os.path.join(ROOT, "tests", "golden", "badsunks", name)
    meta = json.load(open(os.path.join(d, "case.json")))
    argv = [os.path.join(d, n) for n in ("f1", "f2", "s1", "s2")] + [str(tmp_path / "out")]
    if meta["rc"] != 0:
        with pytest.raises(SystemExit):
            badsunks.main(argv)
        return
    capsys.readouterr()
    assert badsunks.main(argv) == 0
    out = capsys.readouterr().out
    assert (tmp_path / "out").read_text() == open(os.path.join(d, "expected")).read()
    ref = open(os.path.join(d, "stdout")).read().splitlines()
    assert [ln for ln in out.splitlines() if ln.startswith("mean coverage")] == [ln for ln in ref if ln.startswith("mean coverage")]
    assert out.splitlines()[-1] == ref[-1] == str(meta["n_bad"])
