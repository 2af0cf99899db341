"""BASELINE.json configs[4] (SUNK lengths other than 20: k = 21 and k = 25) at a size that only fits HBM: a 200 Mb surrogate
in 24 contigs, its exact SUNK set (~1e8 keys) in the minimiser-addressed table, 30 000 synthetic ONT reads (~1.2e9 bases) --
the workload of profiles/r02/c_k2*_1e8keys_bench.json, generated on the GPU by the same code as bench.py's.

Checked like config 3 (tests/test_config3_gpu.py): slices of the stream against the oracle holding the SUNKs that occur in
them (head; middle, started from the carry the GPU reports for everything before it; tail), counters, stream order, no row
repeating its predecessor's location group (kmerpos_annot3.nim:82-96), two passes with identical bytes, and device batches of
uneven sizes with `prevLoc` threaded through."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from gavisunk_amd import _lib, engine, synth_gpu
from oracle import oracle

pytestmark = pytest.mark.gpu

GENOME_MB, READS, SEED = 200.0, 30_000, 30


def _pass(table, reads, carry=_lib.GS_NONE):
    L = _lib.lib()
    n, c_out, ctr = C.c_uint64(0), C.c_uint32(0), _lib.GsCounters()
    _lib.check(L.gs_tag_launch(table._h, reads._h, carry))
    _lib.check(L.gs_tag_finish(table._h, reads._h, None, 0, C.byref(n), C.byref(c_out), C.byref(ctr)))
    vp = C.c_void_p()
    _lib.check(L.gs_tag_hits_view(table._h, reads._h, C.byref(vp), None))
    return engine._view(vp.value, n.value, engine.HIT_DTYPE).copy(), c_out.value, ctr.as_dict()


@pytest.mark.parametrize("k", [21, 25])
def test_other_sunk_lengths_at_1e8_keys(k):
    gen = synth_gpu.Genome(int(GENOME_MB * 1e6), n_contigs=24, seed=SEED)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, k)
    table = sunks.table()
    rng = np.random.default_rng([SEED, 0])
    rs = synth_gpu.ReadSet(gen, synth_gpu.lognormal_lengths(rng, READS, median=32000.0), seed=SEED * 1000)
    gen.close()
    reads = engine.device_reads_from_device_ascii(rs.ptr, rs.off, k)
    L = _lib.lib()
    try:
        assert table.layout.startswith("minimizer-hbm") and f"mod-minimiser of length {k - 4} " in table.layout
        assert sunks.n > 80_000_000 and rs.n_bases > 1e9
        whole, carry, c = _pass(table, reads)
        rlen = (rs.off[1:] - rs.off[:-1]).astype(np.int64)
        nwin = np.maximum(rlen - k + 1, 1)
        assert c["windows"] == int(nwin.sum()) and len(whole) == c["emitted"] <= c["raw_hits"] <= c["windows"]
        assert 0.1 < c["raw_hits"] / c["windows"] < 0.4 and len(whole) > 10_000 and c["fallback_tiles"] < 0.05 * c["tiles"]
        key = whole["read"].astype(np.int64) << 32 | whole["pos"].astype(np.int64)
        assert np.all(np.diff(key) > 0) and np.all(whole["pos"] < nwin[whole["read"]])
        eloc = synth_gpu.gather(0, L.gs_sunkset_dev_loc(sunks._h), np.uint32, whole["payload"])
        assert np.all(eloc[1:] != eloc[:-1]) and carry != _lib.GS_NONE
        again, carry2, c2 = _pass(table, reads)
        assert carry2 == carry and hashlib.sha256(again.tobytes()).digest() == hashlib.sha256(whole.tobytes()).digest()
        # uneven device batches
        cuts, parts, cr = [0, 1, 9_000, 9_001, 29_000, READS], [], _lib.GS_NONE
        for a, b in zip(cuts, cuts[1:]):
            dr = engine.device_reads_from_device_ascii(rs.ptr + int(rs.off[a]), rs.off[a:b + 1] - rs.off[a], k)
            h, cr, _ = _pass(table, dr, cr)
            dr.close()
            h["read"] += np.uint32(a)
            parts.append(h)
        assert cr == carry and np.array_equal(np.concatenate(parts), whole)
        # slices against the oracle
        for lo, hi in ((0, 300), (15_000, 15_200), (READS - 150, READS)):
            prev = _lib.GS_NONE
            if lo:
                dr = engine.device_reads_from_device_ascii(rs.ptr, rs.off[: lo + 1], k)
                _, prev, _ = _pass(table, dr)
                dr.close()
            seqs = rs.host_reads(lo, hi)
            q = np.unique(np.concatenate([oracle.canonical_windows(s if len(s) >= k else s + b"A" * (k - len(s)), k) for s in seqs]))
            rows = synth_gpu.select_rows(0, L.gs_sunkset_dev_code(sunks._h), sunks.n, q)
            code = synth_gpu.gather(0, L.gs_sunkset_dev_code(sunks._h), np.uint64, rows)
            pos = synth_gpu.gather(0, L.gs_sunkset_dev_pos(sunks._h), np.uint64, rows)
            grp = synth_gpu.gather(0, L.gs_sunkset_dev_group(sunks._h), np.uint32, rows)
            loc = synth_gpu.gather(0, L.gs_sunkset_dev_loc(sunks._h), np.uint32, rows)
            t = oracle.Table(k, code, code, (pos >> np.uint64(32)).astype(np.uint32), grp.astype(np.int64))
            w = np.nonzero(loc == prev)[0] if prev != _lib.GS_NONE else []
            buf, off = oracle.concat_reads(seqs)
            rc, oh, _, _ = t.tag(buf, off, prev_row=int(w[0]) if len(w) else oracle.NONE, cap=1 << 24)
            mine = whole[(whole["read"] >= lo) & (whole["read"] < hi)]
            assert rc == 0 and len(mine) == len(oh) > 300
            assert np.array_equal(mine["read"] - lo, oh["read"]) and np.array_equal(mine["pos"], oh["pos"])
            assert np.array_equal(mine["payload"], rows[oh["row"]])
    finally:
        reads.close()
        table.close()
        sunks.close()
        rs.close()
