"""The whole hot path through the drop-in command lines, file to file, in the order of workflow/rules/tagONT.smk
(plus the GPU SUNK definition in front): every intermediate file is compared with the oracle's."""
import io

import numpy as np
import pytest

from gavisunk_amd import (badsunks, define_sunks, diag_filter_step2, diag_filter_v3, kmerpos_annot3, process_by_contig,
                          rlen, synth)
from oracle import oracle, sunkdef

pytestmark = pytest.mark.gpu


def test_file_to_file_pipeline(tmp_path):
    rng = np.random.default_rng(42)
    fam = rng.integers(0, 4, 1500, dtype=np.uint8)
    haps = []
    base = rng.integers(0, 4, 120_000, dtype=np.uint8)
    for h in range(2):
        s = base.copy()
        snp = rng.random(len(s)) < 0.002
        s[snp] = rng.integers(0, 4, int(snp.sum()), dtype=np.uint8)
        for _ in range(6):
            p = int(rng.integers(0, len(s) - 1500))
            s[p:p + 1500] = fam
        haps.append(s)
    d = tmp_path
    for h, s in enumerate(haps, 1):
        (d / f"hap{h}.fa").write_bytes(b">hap%d_ctg\n" % h + synth.codes_to_ascii(s) + b"\n")
        (d / f"hap{h}.fa.fai").write_text(f"hap{h}_ctg\t{len(s)}\t10\t{len(s)}\t{len(s) + 1}\n")
    # SUNK definition (GPU) -> db + loc, checked against the NumPy restatement
    assert define_sunks.main(["-k", "20", str(d / "hap1.fa"), str(d / "hap2.fa"), "--db", str(d / "db"), "--loc", str(d / "loc")]) == 0
    exp = sunkdef.define_sunks([("hap1_ctg", synth.codes_to_ascii(haps[0])), ("hap2_ctg", synth.codes_to_ascii(haps[1]))], 20)
    exp.write_loc(str(d / "loc.exp"))
    assert (d / "loc").read_text() == (d / "loc.exp").read_text() and len(exp) > 1000
    sunkpos = {}
    for h, s in enumerate(haps, 1):
        lens = synth.lognormal_lengths(rng, 60, median=15000, sigma=0.5, lo=2000, hi=60000)
        reads = synth.simulate_reads(rng, [s], lens)
        synth.write_fasta(str(d / f"r{h}.fq"), [f"h{h}_read{i}" for i in range(len(reads))], reads, fastq=True)
        # tag + read lengths
        assert kmerpos_annot3.main([str(d / f"r{h}.fq"), str(d / "db"), str(d / "loc"), str(d / f"h{h}.sunkpos")]) == 0
        rc, txt = oracle.tag_files(str(d / f"r{h}.fq"), str(d / "db"), str(d / "loc"))
        assert rc == 0 and (d / f"h{h}.sunkpos").read_text() == txt and txt.count("\n") > 50
        assert rlen.main([str(d / f"r{h}.fq"), str(d / f"h{h}.rlen")]) == 0
        assert (d / f"h{h}.rlen").read_text() == oracle.read_lengths(str(d / f"r{h}.fq"))
        # diagonal filter, both steps
        buf = io.StringIO()
        assert diag_filter_v3.run(str(d / f"h{h}.sunkpos"), str(d / f"hap{h}.fa.fai"), out=buf) == 0
        diag = oracle.diag_filter_v3(oracle.parse_sunkpos(txt), [f"hap{h}_ctg"])
        assert buf.getvalue() == diag
        (d / f"h{h}.diag").write_text(diag)
        buf2 = io.StringIO()
        assert diag_filter_step2.run(str(d / f"h{h}.sunkpos"), str(d / f"h{h}.diag"), out=buf2) == 0
        assert buf2.getvalue() == oracle.diag_filter_step2(txt, diag)
        (d / f"h{h}.diag2").write_text(buf2.getvalue())
        sunkpos[h] = buf2.getvalue()
    # bad SUNKs over both haplotypes
    assert badsunks.main([str(d / "hap1.fa.fai"), str(d / "hap2.fa.fai"), str(d / "h1.diag2"), str(d / "h2.diag2"), str(d / "bad")]) == 0
    ebad = oracle.badsunks((d / "hap1.fa.fai").read_text(), (d / "hap2.fa.fai").read_text(), sunkpos[1], sunkpos[2])
    assert set((d / "bad").read_text().split()) == ebad
    # per-contig validation
    for h in (1, 2):
        assert process_by_contig.main([str(d / "loc"), str(d / f"h{h}.diag2"), str(d / f"h{h}.rlen"), str(d / "bad"),
                                       str(d / f"h{h}.tsv"), str(d / f"h{h}.bed")]) == 0
        etsv, ebed = oracle.process_by_contig(sunkpos[h], (d / f"h{h}.rlen").read_text(), (d / "bad").read_text())
        assert (d / f"h{h}.tsv").read_text() == etsv and etsv.count("\n") > 10
        assert ((d / f"h{h}.bed").read_text() if (d / f"h{h}.bed").exists() else "") == ebed
