"""BASELINE.json configs[2] ("config 3") at FULL size: CHM13-scale haplotype surrogate (24 contigs, 3.1 Gb), its exact SUNK
set (~1.5e9 20-mers, gs_sunks_define_device) in an HBM-resident minimiser-addressed table, 1 000 000 synthetic ONT reads
(~4e10 bases) -- the workload bench.py's headline line is measured on, generated on the GPU by the same code.

The oracle cannot hold a 1.5e9-key table or walk 4e10 bases in a test, so the full-size run is checked through
* slices of the stream against the oracle (the head of the stream, and a slice in its middle started from the carry the
  GPU itself reports for everything before it), the oracle's table holding exactly the SUNKs that occur in the slice;
* properties that do not depend on size: counters add up, the emitted list is in stream order and inside every read, no row
  repeats the location group of its predecessor (kmerpos_annot3.nim:82-96) across read boundaries too, two passes give
  identical bytes, and cutting the read set into device batches with `prevLoc` threaded through changes nothing.
"""
import ctypes as C
import hashlib

import numpy as np
import pytest

from gavisunk_amd import _lib, engine, synth_gpu
from oracle import oracle

pytestmark = pytest.mark.gpu

K = 20
GENOME_MB = 3100.0
READS = 1_000_000
SEED = 30


@pytest.fixture(scope="module")
def cfg3():
    gen = synth_gpu.Genome(int(GENOME_MB * 1e6), n_contigs=24, seed=SEED)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, K)
    table = sunks.table()
    rng = np.random.default_rng([SEED, 0])
    rs = synth_gpu.ReadSet(gen, synth_gpu.lognormal_lengths(rng, READS, median=32000.0), seed=SEED * 1000)
    gen.close()
    reads = engine.device_reads_from_device_ascii(rs.ptr, rs.off, K)
    yield {"sunks": sunks, "table": table, "rs": rs, "reads": reads}
    reads.close()
    table.close()
    sunks.close()
    rs.close()


def _pass(table, reads, carry=_lib.GS_NONE):
    """One tagging pass; the hit list as a numpy view of the pinned host mirror (copied: the next pass overwrites it)."""
    L = _lib.lib()
    n, c_out, ctr = C.c_uint64(0), C.c_uint32(0), _lib.GsCounters()
    _lib.check(L.gs_tag_launch(table._h, reads._h, carry))
    _lib.check(L.gs_tag_finish(table._h, reads._h, None, 0, C.byref(n), C.byref(c_out), C.byref(ctr)))
    vp = C.c_void_p()
    _lib.check(L.gs_tag_hits_view(table._h, reads._h, C.byref(vp), None))
    return engine._view(vp.value, n.value, engine.HIT_DTYPE).copy(), c_out.value, ctr.as_dict()


def _oracle_slice(cfg3, lo, hi, prev_loc):
    """Oracle over reads [lo, hi) with the sub-table of SUNKs that occur in them -> (hits, global rows of the sub-table)."""
    sunks, rs = cfg3["sunks"], cfg3["rs"]
    seqs = rs.host_reads(lo, hi)
    q = np.unique(np.concatenate([oracle.canonical_windows(s if len(s) >= K else s + b"A" * (K - len(s)), K) for s in seqs]))
    L = _lib.lib()
    rows = synth_gpu.select_rows(0, L.gs_sunkset_dev_code(sunks._h), sunks.n, q)
    code = synth_gpu.gather(0, L.gs_sunkset_dev_code(sunks._h), np.uint64, rows)
    pos = synth_gpu.gather(0, L.gs_sunkset_dev_pos(sunks._h), np.uint64, rows)
    grp = synth_gpu.gather(0, L.gs_sunkset_dev_group(sunks._h), np.uint32, rows)
    loc = synth_gpu.gather(0, L.gs_sunkset_dev_loc(sunks._h), np.uint32, rows)
    t = oracle.Table(K, code, code, (pos >> np.uint64(32)).astype(np.uint32), grp.astype(np.int64))
    prev_row = oracle.NONE
    if prev_loc != _lib.GS_NONE:  # any row of that location group: the oracle compares (contig, group)
        w = np.nonzero(loc == prev_loc)[0]
        if len(w):
            prev_row = int(w[0])
        else:  # the carried group has no SUNK in this slice: it cannot suppress anything here
            prev_row = oracle.NONE
    buf, off = oracle.concat_reads(seqs)
    rc, oh, _, octr = t.tag(buf, off, prev_row=prev_row, cap=1 << 24)
    assert rc == 0
    return oh, rows, octr


def test_config3_full_size(cfg3):
    table, reads, rs, sunks = cfg3["table"], cfg3["reads"], cfg3["rs"], cfg3["sunks"]
    assert table.layout.startswith("minimizer-hbm") and sunks.n > 1_000_000_000 and rs.n_bases > 3.5e10
    whole, carry, c = _pass(table, reads)
    rlen = (rs.off[1:] - rs.off[:-1]).astype(np.int64)
    nwin = np.maximum(rlen - K + 1, 1)
    # counters
    assert c["windows"] == int(nwin.sum()) and len(whole) == c["emitted"] <= c["raw_hits"] <= c["windows"]
    assert 0.1 < c["raw_hits"] / c["windows"] < 0.4 and len(whole) > 1_000_000
    assert c["fallback_tiles"] < 0.05 * c["tiles"]  # tiles the fast kernel hands to its exact form (LDS budgets)
    # stream order, positions inside the read (pos is deflated, so it is below the window count all the more)
    key = whole["read"].astype(np.int64) << 32 | whole["pos"].astype(np.int64)
    assert np.all(np.diff(key) > 0)
    assert np.all(whole["pos"] < nwin[whole["read"]])
    # never the location group of the previous emitted row, reads notwithstanding
    L = _lib.lib()
    eloc = np.concatenate([synth_gpu.gather(0, L.gs_sunkset_dev_loc(sunks._h), np.uint32, whole["payload"][a:a + (1 << 24)])
                           for a in range(0, len(whole), 1 << 24)])
    assert np.all(eloc[1:] != eloc[:-1])
    assert carry != _lib.GS_NONE and carry < sunks.n_groups

    # a second pass gives the same bytes
    again, carry2, c2 = _pass(table, reads)
    assert carry2 == carry and c2["raw_hits"] == c["raw_hits"]
    assert hashlib.sha256(again.tobytes()).digest() == hashlib.sha256(whole.tobytes()).digest()
    del again

    # device batches of very different sizes, `prevLoc` threaded through
    cuts = [0, 1, 300_000, 300_001, 999_000, READS]
    parts, cr, base, raw = [], _lib.GS_NONE, 0, 0
    for a, b in zip(cuts, cuts[1:]):
        off = rs.off[a:b + 1] - rs.off[a]
        dr = engine.device_reads_from_device_ascii(rs.ptr + int(rs.off[a]), off, K)
        h, cr, cc = _pass(table, dr, cr)
        dr.close()
        h["read"] += np.uint32(a)
        parts.append(h)
        raw += cc["raw_hits"]
    parts = np.concatenate(parts)
    assert cr == carry and raw == c["raw_hits"] and np.array_equal(parts, whole)
    del parts

    # slices of the stream against the oracle
    for lo, hi in ((0, 600), (500_000, 500_300), (READS - 200, READS)):
        prev = _lib.GS_NONE
        if lo:
            off = rs.off[: lo + 1]
            dr = engine.device_reads_from_device_ascii(rs.ptr, off, K)
            _, prev, _ = _pass(table, dr)
            dr.close()
        oh, rows, octr = _oracle_slice(cfg3, lo, hi, prev)
        mine = whole[(whole["read"] >= lo) & (whole["read"] < hi)]
        assert len(mine) == len(oh) > 1000
        assert np.array_equal(mine["read"] - lo, oh["read"]) and np.array_equal(mine["pos"], oh["pos"])
        assert np.array_equal(mine["payload"], rows[oh["row"]])
