"""A bounded, seeded slice of the fuzzers under tools/ (fuzz_tag.py, fuzz_ingest.py) inside the GPU suite: every round tags one
random (genome, SUNK table, read set) under EVERY table layout -- the default (place records / per-key slots by k), compact-hbm,
minimizer, minimizer-exact, minimizer-slots, generic -- and compares hit arrays, carries and counters bit for bit, small rounds
against the CPU oracle too; hot regions forced on, off and left to the library's rule in turn.  The seeds are fixed (drawn once
from the fuzzer's own generator, kept to rounds of < 4e6 bases) so that the suite stays deterministic and short."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# seed: k of the round (what tools/fuzz_tag.draw gives) -- k = 11 .. 32, tables of 2e2 .. 4e5 keys, reads of 0 .. 3.5e6 bases
TAG_SEEDS = [40000, 40001, 40002, 40003, 40005, 40006, 40007, 40008, 40011, 40012, 40013, 40015, 40016, 40017, 40019, 40020, 40021,
             40023, 40024, 40026, 40029, 40030, 40032, 40033, 40035, 40039, 40041, 40043, 40044, 40045, 40046, 40047, 40048, 40051,
             40055, 40057, 40058, 40060, 40062, 40063, 40066, 40067, 40068, 40069, 40070, 40073, 40075, 40078, 40079, 40082, 40083,
             40085, 40086, 40089, 40090, 40091, 40092, 40094, 40095, 40099, 40101, 40103, 40104, 40105, 40106, 40109, 40112, 40115,
             40116, 40118, 40119, 40120, 40121, 40126]


def test_fuzz_tag_rounds_every_layout():
    import fuzz_tag

    seen, ks = {}, set()
    for i, seed in enumerate(TAG_SEEDS):
        ok, msg, names = fuzz_tag.one_round(seed, hot=(None, 0, 3)[i % 3])
        assert ok, msg
        ks.add(fuzz_tag.K)
        for n, v in names.items():
            seen[n] = seen.get(n, 0) + v
    assert len(TAG_SEEDS) >= 60 and {11, 16, 17, 18, 19, 20, 21, 22, 23, 25, 28, 31, 32} <= ks
    assert {"minimizer-hbm", "compact-hbm", "generic"} <= set(seen), seen  # (minimizer-exact reports itself as minimizer-hbm)


def test_fuzz_ingest_files(tmp_path):
    import fuzz_ingest

    path = str(tmp_path / "f")
    counts = {"ok": 0, "refused": 0, "empty": 0}
    for seed in range(7000, 7060):
        what, _ = fuzz_ingest.one_file(seed, path)
        counts[what] += 1
    assert counts["ok"] >= 40, counts


def test_fuzz_filter_rounds(tmp_path):
    """tools/fuzz_filters.py: the file-level drop-ins (diag_filter_v3 + step 2, process-by-contig, badsunks: all users of the
    device chain) and the SUNK definition against the oracle, 25 seeded rounds of each"""
    import fuzz_filters

    for seed in range(92000, 92025):
        for name, fn in (("diag", fuzz_filters.fuzz_diag), ("pair", fuzz_filters.fuzz_pair), ("sunkdef", fuzz_filters.fuzz_sunkdef),
                         ("badsunks", fuzz_filters.fuzz_badsunks)):
            err = fn(seed, str(tmp_path))
            assert err is None, (name, seed, err)
