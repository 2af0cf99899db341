"""Inter-SUNK distance stage against PROVISIONAL goldens: outputs of the reference's own
process-by-contig_lowmem_AR.py run in the build container under pandas 2.x with a graph_tool stand-in
(tests/golden/make_golden_pair.py, tests/golden/_shim/).  CPU: the oracle restatement; GPU: the drop-in stub."""
import json
import os

import pytest

from conftest import GOLDEN
from oracle import oracle

CASES = ["pair_amy", "pair_multi_k20", "pair_multi_k25", "pair_synth"]


def _items(case):
    d = os.path.join(GOLDEN, case)
    meta = json.load(open(os.path.join(d, "case.json")))
    for c in meta["contigs"]:
        assert c["rc"] == 0
        yield d, c


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_script(case):
    n = 0
    for d, c in _items(case):
        t = c["tag"]
        tsv, bed = oracle.process_by_contig(open(f"{d}/{t}.sunkpos").read(), open(f"{d}/rlen").read(), open(f"{d}/{t}.bad").read())
        assert tsv == open(f"{d}/{t}.expected.tsv").read()
        assert bed == (open(f"{d}/{t}.expected.bed").read() if c["has_bed"] else "")
        n += tsv.count("\n")
    assert n > (500 if case == "pair_synth" else 5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_stub_matches_reference_script(case, tmp_path):
    from gavisunk_amd import process_by_contig

    for d, c in _items(case):
        t = c["tag"]
        tsv, bed = tmp_path / f"{t}.tsv", tmp_path / f"{t}.bed"
        assert process_by_contig.main([f"{d}/{t}.loc", f"{d}/{t}.sunkpos", f"{d}/rlen", f"{d}/{t}.bad", str(tsv), str(bed)]) == 0
        assert tsv.read_text() == open(f"{d}/{t}.expected.tsv").read()
        assert (bed.read_text() if bed.exists() else "") == (open(f"{d}/{t}.expected.bed").read() if c["has_bed"] else "")
