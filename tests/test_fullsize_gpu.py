"""BASELINE.json configs[1] at full size (5 Mb contig, 10 000 synthetic ONT reads, 242 Mbase, k = 20).

The whole emitted list (all 10 000 reads, 37 280 rows) is compared with what the reference's own `kmerpos_annot3` binary
printed on the same inputs (tests/golden/config2_full/c2.sunkpos.gz, made by tests/golden/make_golden_config2.py in the
build container; tests/test_oracle_golden.py holds the oracle to the same file on the CPU).  On top of that, properties
that do not depend on size:

* the three independent probe paths (LDS join, L2-partitioned table, HBM-resident compact table) return the same array, bit for bit;
* splitting the read set into device batches with `prevLoc` threaded through them changes nothing;
* the emitted list is in stream order, inside every read, and never repeats the location group of its predecessor
  (kmerpos_annot3.nim:82-96: that is what the `continue` is for), across read boundaries too;
* counters add up (windows = sum of per-read windows; emitted <= raw hits);
* two slices in the middle of the stream equal the oracle's output for those reads, started from the carry the GPU itself
  reports for everything before them.
"""
import numpy as np
import pytest

from gavisunk_amd import _lib, engine, synth
from oracle import oracle, sunkdef

pytestmark = pytest.mark.gpu

K = 20
CONFIG2_ROWS = {}


@pytest.fixture(scope="module")
def config2():
    seq, sunks, names, reads = synth.config2(sunkdef.define_sunks, seed=20, n_reads=10000, k=K)
    keys = sunks.code.astype(np.uint64)
    _, loc = np.unique(sunks.group, return_inverse=True)  # one contig: the group decides the location id
    loc = loc.astype(np.uint32)
    order = np.argsort(keys, kind="stable")
    seqs = [synth.codes_to_ascii(r) for r in reads]
    CONFIG2_ROWS["start"], CONFIG2_ROWS["group"] = sunks.start[order].tolist(), sunks.group[order].tolist()
    return keys[order], loc[order], seqs


def _tag(monkeypatch, layout, keys, loc, batches, carry=_lib.GS_NONE):
    if layout == "auto":
        monkeypatch.delenv("GAVISUNK_TABLE", raising=False)
    else:
        monkeypatch.setenv("GAVISUNK_TABLE", layout)
    tab = engine.DeviceTable(K, keys, loc, np.arange(len(keys), dtype=np.uint32))
    parts, ctrs, base = [], [], 0
    for seqs in batches:
        batch = engine.ReadBatch.from_ascii(seqs)
        dev = engine.DeviceReads(batch, K)
        batch.close()
        st, h, carry, ctr = engine.tag(tab, dev, carry)
        assert st == 0
        h = h.copy()
        h["read"] += base
        base += len(seqs)
        parts.append(h)
        ctrs.append(dict(ctr))
        dev.close()
    name = tab.layout.split(":")[0]
    tab.close()
    return np.concatenate(parts), carry, ctrs, name


def test_config2_full_size_properties(config2, monkeypatch):
    keys, loc, seqs = config2
    whole, carry, ctrs, name = _tag(monkeypatch, "auto", keys, loc, [seqs])
    assert name == "minimizer-hbm"  # (since round 2 the default for every k >= 16 table, this small one included)
    c = ctrs[0]
    nwin = np.array([max(len(s) - K + 1, 1) for s in seqs], dtype=np.int64)
    assert c["windows"] == int(nwin.sum()) > 240_000_000 and len(whole) == c["emitted"] <= c["raw_hits"]
    assert 0.2 < c["raw_hits"] / c["windows"] < 0.3 and len(whole) > 30_000

    # every row equals the reference binary's output on these inputs
    import gzip
    import os

    from conftest import GOLDEN
    want = gzip.open(os.path.join(GOLDEN, "config2_full", "c2.sunkpos.gz"), "rb").read().split(b"\n")[:-1]
    start, group = CONFIG2_ROWS["start"], CONFIG2_ROWS["group"]
    got = [b"r%d\t%d\tctg5M\t%d\t%d" % (a, b, start[p], group[p])
           for a, b, p in zip(whole["read"].tolist(), whole["pos"].tolist(), whole["payload"].tolist())]
    assert got == want

    # stream order, positions inside the read (pos is deflated, so it is below the window count all the more)
    key = whole["read"].astype(np.int64) << 32 | whole["pos"].astype(np.int64)
    assert np.all(np.diff(key) > 0)
    assert np.all(whole["pos"] < nwin[whole["read"]])
    # never the location of the previous emitted row, reads notwithstanding; the carry out is a location of the table
    eloc = loc[whole["payload"]]
    assert np.all(eloc[1:] != eloc[:-1])
    assert carry != _lib.GS_NONE and carry <= loc.max()

    # the other probe paths agree bit for bit
    for layout, want in (("compact-hbm", "compact-hbm"), ("generic", "generic")):
        other, carry2, ctrs2, name2 = _tag(monkeypatch, layout, keys, loc, [seqs])
        assert name2 == want
        assert np.array_equal(other, whole) and carry2 == carry and ctrs2[0]["raw_hits"] == c["raw_hits"]

    # device batches of very different sizes, `prevLoc` threaded through
    cuts = [0, 1, 2, 1500, 1501, 6000, 10000]
    parts, carry3, ctrs3, _ = _tag(monkeypatch, "auto", keys, loc, [seqs[a:b] for a, b in zip(cuts, cuts[1:])])
    assert np.array_equal(parts, whole) and carry3 == carry
    assert sum(x["raw_hits"] for x in ctrs3) == c["raw_hits"] and sum(x["windows"] for x in ctrs3) == c["windows"]

    # two slices of the stream against the oracle, each started from the GPU's own carry for everything before it
    t = oracle.Table(K, keys, keys, np.zeros(len(keys), np.uint32), loc.astype(np.int64))
    row_of_loc = {int(l): i for i, l in enumerate(loc)}  # any row with that location: the oracle compares locations
    for lo, hi in ((4000, 4060), (9950, 10000)):
        _, carry_before, _, _ = _tag(monkeypatch, "auto", keys, loc, [seqs[:lo]])
        buf, off = oracle.concat_reads(seqs[lo:hi])
        rc, oh, _, _ = t.tag(buf, off, prev_row=row_of_loc[int(carry_before)])
        assert rc == 0
        mine = whole[(whole["read"] >= lo) & (whole["read"] < hi)]
        assert len(mine) == len(oh) > 50
        assert np.array_equal(mine["read"] - lo, oh["read"]) and np.array_equal(mine["pos"], oh["pos"])
        assert np.array_equal(mine["payload"], oh["row"])
