"""Inter-SUNK distance stage against PROVISIONAL goldens: outputs of the reference's own
process-by-contig_lowmem_AR.py run in the build container under pandas 2.x with a graph_tool stand-in
(tests/golden/make_golden_pair.py, tests/golden/_shim/).  CPU: the oracle restatement; GPU: the drop-in stub."""
import json
import os

import pytest

from conftest import GOLDEN
from oracle import oracle

CASES = ["pair_amy", "pair_multi_k20", "pair_multi_k25", "pair_synth", "pair_long"]


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
    assert n > (500 if case in ("pair_synth", "pair_long") else 5)


def test_array_form_of_the_oracle_equals_the_loop_form():
    """oracle.pair_component_read_np (NumPy / SciPy; what the GPU tests use for reads with thousands of hits) against the
    line-by-line restatement, on random reads with many multi-position IDs and `value_counts` ties."""
    import numpy as np

    rng = np.random.default_rng(7)
    for _ in range(200):
        n = int(rng.integers(2, 60))
        p = np.sort(rng.choice(np.arange(1, 5000), size=n, replace=False))
        s = 100000 + p + rng.integers(-40, 40, n)
        jump = rng.random(n) < 0.15
        s[jump] = rng.integers(0, 200000, int(jump.sum()))
        order = np.argsort(s, kind="stable")
        p, s = p[order], s[order]
        g = (s // int(rng.choice([50, 200, 2000]))) * 50
        assert oracle.pair_component_read(g.tolist(), p.tolist(), s.tolist()) == oracle.pair_component_read_np(g.tolist(), p.tolist(), s.tolist())


def test_tie_census_of_the_goldens():
    """Which goldens can contain a decision of the graph_tool stand-in / of pandas 2.x (tests/golden/tie_census.py): the AMY
    locus, the two multi-contig cases and the long-read case (10^2 .. 10^3 hits per read, ~200 multi-position IDs) have no
    `value_counts` tie and no largest-component tie -- whatever rule the pinned pandas 1.3.4 / graph-tool 2.43 apply, it
    never had to choose there.  pair_synth does exercise `value_counts` ties (first-seen order, pandas 2.x): it pins the
    clean-up code path only under that rule."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("tie_census", os.path.join(GOLDEN, "tie_census.py"))
    tc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tc)
    res = {c: tc.census_case(c) for c in CASES}
    for c in ("pair_amy", "pair_multi_k20", "pair_multi_k25", "pair_long"):
        assert res[c]["value_count_ties"] == 0 and res[c]["largest_ties"] == 0, (c, res[c])
    assert res["pair_long"]["max_hits"] >= 1000 and res["pair_long"]["multipos_ids"] > 100
    assert res["pair_synth"]["largest_ties"] == 0 and res["pair_synth"]["value_count_ties"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [None, "2000"])
@pytest.mark.parametrize("case", CASES)
def test_gpu_stub_matches_reference_script(case, chunk, tmp_path, monkeypatch):
    """chunk: the pair kernels take the reads in chunks of at most that many candidate pairs (a genome-scale hit list has 1e10 of
    them): the goldens once in one chunk, once in dozens."""
    from gavisunk_amd import process_by_contig

    if chunk:
        monkeypatch.setenv("GAVISUNK_PAIR_CHUNK", chunk)

    for d, c in _items(case):
        t = c["tag"]
        tsv, bed = tmp_path / f"{t}.tsv", tmp_path / f"{t}.bed"
        assert process_by_contig.main([f"{d}/{t}.loc", f"{d}/{t}.sunkpos", f"{d}/rlen", f"{d}/{t}.bad", str(tsv), str(bed)]) == 0
        assert tsv.read_text() == open(f"{d}/{t}.expected.tsv").read()
        assert (bed.read_text() if bed.exists() else "") == (open(f"{d}/{t}.expected.bed").read() if c["has_bed"] else "")
