"""Diagonal filter on the GPU (gs_diag_filter via the CLI stubs) against the outputs of the
reference's own diag_filter_v3 / diag_filter_step2 binaries, and against the oracle."""
import io

import numpy as np
import pytest

from conftest import DIAG_CASES
from gavisunk_amd import diag_filter_step2, diag_filter_v3, engine
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", DIAG_CASES)
def test_cli_matches_reference_binaries(golden, name):
    c = golden(name)
    for run in c.meta["diag"]:
        buf = io.StringIO()
        assert diag_filter_v3.run(c.path(run["sunkpos"]), c.path(run["fai"]), out=buf) == 0
        assert buf.getvalue() == c.text(run["out"])
        buf2 = io.StringIO()
        assert diag_filter_step2.run(c.path(run["sunkpos"]), c.path(run["out"]), out=buf2) == 0
        assert buf2.getvalue() == c.text(run["out2"])


@pytest.mark.parametrize("name", ["tie", "amy", "multi_k20"])
def test_keep_mask_equals_step2(golden, name):
    c = golden(name)
    run = c.meta["diag"][0]
    sp = engine.SunkPos(c.path(run["sunkpos"]))
    fai = [ln.split("\t")[0] for ln in c.text(run["fai"]).splitlines() if ln]
    bc, bg, bd, keep = engine.diag_filter(sp, fai)
    kept = b"".join(sp.row_text(i) + b"\n" for i in np.nonzero(keep)[0]).decode()
    # unique read names in these fixtures: the in-memory mask and the by-name file join agree
    assert kept == c.text(run["out2"])


def test_random_rows_match_oracle(tmp_path):
    rng = np.random.default_rng(5)
    contigs = [f"ctg{i}" for i in range(12)] + ["notinfai1", "notinfai2"]
    rows = []
    for r in range(300):
        n = int(rng.integers(1, 40))
        picks = rng.choice(len(contigs), size=int(rng.integers(1, 5)), replace=False)
        base = int(rng.integers(0, 10**6))
        pos = 0
        for _ in range(n):
            pos += int(rng.integers(1, 3000))
            c = contigs[int(rng.choice(picks))]
            jitter = int(rng.integers(-3000, 3000)) if rng.random() < 0.3 else int(rng.integers(-50, 50))
            sign = 1 if r % 2 else -1
            start = base + sign * pos + jitter + (10**7 if sign < 0 else 0)
            rows.append((f"r{r}", pos, c, start, (start // 500) * 500))
    text = "".join(f"{a}\t{b}\t{c}\t{d}\t{e}\n" for a, b, c, d, e in rows)
    p = tmp_path / "x.sunkpos"
    p.write_text(text)
    fai = tmp_path / "x.fai"
    fai.write_text("".join(f"{c}\t1\n" for c in contigs[:12]))
    buf = io.StringIO()
    assert diag_filter_v3.run(str(p), str(fai), out=buf) == 0
    exp = oracle.diag_filter_v3(oracle.parse_sunkpos(text), contigs[:12])
    assert buf.getvalue() == exp and exp.count("\n") > 50


def test_coordinates_outside_uint32_take_the_wide_path(tmp_path):
    """diag_filter_v3.nim:148-150 parses pos / start / ID with parseInt: negative and > 2^32 values are legal input of the filter
    (found by tools/fuzz_filters.py when the CLI moved to the uint32 row set)"""
    rows = []
    for r in range(40):
        base = -2000 + 100 * r if r % 2 else (1 << 33) + 1000 * r
        for i in range(12):
            s = base + 300 * i + (7 if i % 5 else 2600)
            rows.append((f"w{r}", 300 * i, "ctgA" if i % 4 else "ctgB", s, (s // 500) * 500))
    text = "".join(f"{a}\t{b}\t{c}\t{d}\t{e}\n" for a, b, c, d, e in rows)
    p = tmp_path / "w.sunkpos"
    p.write_text(text)
    fai = tmp_path / "w.fai"
    fai.write_text("ctgA\t1\nctgB\t1\n")
    buf = io.StringIO()
    assert diag_filter_v3.run(str(p), str(fai), out=buf) == 0
    exp = oracle.diag_filter_v3(oracle.parse_sunkpos(text), ["ctgA", "ctgB"])
    assert buf.getvalue() == exp and exp.count("\n") >= 30
