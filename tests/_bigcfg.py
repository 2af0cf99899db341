"""Shared checks of the full-size GPU tests (BASELINE.json configs[2], [3], [4]: tests/test_config3_gpu.py,
test_config_diploid_gpu.py, test_config4_gpu.py).

A genome-scale run cannot be walked by the oracle as a whole (a 1e9-key table, 1e10 .. 4e10 bases), so it is checked through
* slices of the stream against the oracle (the head of the stream, slices in its middle started from the carry the GPU itself
  reports for everything before them, the tail), the oracle's table holding exactly the SUNKs that occur in the slice (found
  by a plain set-membership kernel of the bench-support library: independent of the product's tables);
* properties that do not depend on size: counters add up, the emitted list is in stream order and inside every read, no row
  repeats the location group of its predecessor (kmerpos_annot3.nim:82-96) across read boundaries too, two passes give
  identical bytes, and cutting the read set into device batches with `prevLoc` threaded through changes nothing.
"""
import ctypes as C
import hashlib

import numpy as np

from gavisunk_amd import _lib, engine, synth_gpu
from oracle import oracle


def one_pass(table, reads, carry=_lib.GS_NONE):
    """One tagging pass; the hit list as a numpy copy of the pinned host mirror (the next pass overwrites it)."""
    L = _lib.lib()
    n, c_out, ctr = C.c_uint64(0), C.c_uint32(0), _lib.GsCounters()
    _lib.check(L.gs_tag_launch(table._h, reads._h, carry))
    _lib.check(L.gs_tag_finish(table._h, reads._h, None, 0, C.byref(n), C.byref(c_out), C.byref(ctr)))
    vp = C.c_void_p()
    _lib.check(L.gs_tag_hits_view(table._h, reads._h, C.byref(vp), None))
    return engine._view(vp.value, n.value, engine.HIT_DTYPE).copy(), c_out.value, ctr.as_dict()


def _windows_of(seqs, k):
    """distinct canonical window codes of the reads, 250 reads at a time on a few threads (the C helper and NumPy's sort both run
    without the GIL): a slice of thousands of reads has 1e8 windows"""
    from concurrent.futures import ThreadPoolExecutor

    def part(a):
        return np.unique(np.concatenate([oracle.canonical_windows(s if len(s) >= k else s + b"A" * (k - len(s)), k) for s in seqs[a:a + 250]]))

    with ThreadPoolExecutor(8) as pool:
        parts = list(pool.map(part, range(0, len(seqs), 250)))
    return np.unique(np.concatenate(parts))


def oracle_slice_inputs(sunks, rs, k, lo, hi, prev_loc):
    """What the oracle needs for reads [lo, hi): the reads, the sub-table of SUNKs that occur in them (rows of the full set found
    by a set-membership kernel of the bench-support library), the row standing for the carried location group."""
    seqs = rs.host_reads(lo, hi)
    q = _windows_of(seqs, k)
    L = _lib.lib()
    rows = synth_gpu.select_rows(0, L.gs_sunkset_dev_code(sunks._h), sunks.n, q)
    code = synth_gpu.gather(0, L.gs_sunkset_dev_code(sunks._h), np.uint64, rows)
    pos = synth_gpu.gather(0, L.gs_sunkset_dev_pos(sunks._h), np.uint64, rows)
    grp = synth_gpu.gather(0, L.gs_sunkset_dev_group(sunks._h), np.uint32, rows)
    loc = synth_gpu.gather(0, L.gs_sunkset_dev_loc(sunks._h), np.uint32, rows)
    prev_row = oracle.NONE
    if prev_loc != _lib.GS_NONE:  # any row of that location group: the oracle compares (contig, group)
        w = np.nonzero(loc == prev_loc)[0]
        if len(w):  # (a carried group without a SUNK in this slice cannot suppress anything here)
            prev_row = int(w[0])
    return seqs, rows, code, pos, grp, prev_row


def oracle_slice_run(k, inputs):
    """-> (hits, global rows of the sub-table, counters); a C call that releases the GIL: slices run side by side"""
    seqs, rows, code, pos, grp, prev_row = inputs
    t = oracle.Table(k, code, code, (pos >> np.uint64(32)).astype(np.uint32), grp.astype(np.int64))
    buf, off = oracle.concat_reads(seqs)
    rc, oh, _, octr = t.tag(buf, off, prev_row=prev_row, cap=max(1 << 20, int(off[-1]) // 8))
    assert rc == 0
    return oh, rows, octr


def oracle_slice(sunks, rs, k, lo, hi, prev_loc):
    return oracle_slice_run(k, oracle_slice_inputs(sunks, rs, k, lo, hi, prev_loc))


def full_size_checks(table, reads, rs, sunks, k, cuts, slices, min_rows, min_slice_rows, hit_rate=(0.1, 0.4)):
    n_reads = rs.n_reads
    whole, carry, c = one_pass(table, reads)
    rlen = (rs.off[1:] - rs.off[:-1]).astype(np.int64)
    nwin = np.maximum(rlen - k + 1, 1)
    # counters
    assert c["windows"] == int(nwin.sum()) and len(whole) == c["emitted"] <= c["raw_hits"] <= c["windows"]
    assert hit_rate[0] < c["raw_hits"] / c["windows"] < hit_rate[1] and len(whole) > min_rows
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
    del eloc, key

    # a second pass gives the same bytes
    again, carry2, c2 = one_pass(table, reads)
    assert carry2 == carry and c2["raw_hits"] == c["raw_hits"]
    assert hashlib.sha256(again.tobytes()).digest() == hashlib.sha256(whole.tobytes()).digest()
    del again

    # device batches of very different sizes, `prevLoc` threaded through
    parts, cr, raw = [], _lib.GS_NONE, 0
    for a, b in zip(cuts, cuts[1:]):
        dr = engine.device_reads_from_device_ascii(rs.ptr + int(rs.off[a]), rs.off[a:b + 1] - rs.off[a], k)
        h, cr, cc = one_pass(table, dr, cr)
        dr.close()
        h["read"] += np.uint32(a)
        parts.append(h)
        raw += cc["raw_hits"]
    parts = np.concatenate(parts)
    assert cuts[0] == 0 and cuts[-1] == n_reads and cr == carry and raw == c["raw_hits"] and np.array_equal(parts, whole)
    del parts

    # slices of the stream against the oracle (>= 1 % of the reads in all): the GPU-side preparation of every slice first, then
    # the oracle over all of them on a few host threads
    from concurrent.futures import ThreadPoolExecutor

    inputs = []
    for lo, hi in slices:
        prev = _lib.GS_NONE
        if lo:
            dr = engine.device_reads_from_device_ascii(rs.ptr, rs.off[: lo + 1], k)
            _, prev, _ = one_pass(table, dr)
            dr.close()
        inputs.append(oracle_slice_inputs(sunks, rs, k, lo, hi, prev))
    with ThreadPoolExecutor(8) as pool:
        results = list(pool.map(lambda x: oracle_slice_run(k, x), inputs))
    del inputs
    for (lo, hi), (oh, rows, octr) in zip(slices, results):
        mine = whole[(whole["read"] >= lo) & (whole["read"] < hi)]
        assert len(mine) == len(oh) > min_slice_rows
        assert np.array_equal(mine["read"] - lo, oh["read"]) and np.array_equal(mine["pos"], oh["pos"])
        assert np.array_equal(mine["payload"], rows[oh["row"]])
    return whole, carry, c


def chain_slice_check(table, reads, rs, sunks, contig_names, n_slice, k):
    """The device-resident second half (csrc/gs_chain.hip) over the WHOLE hit list of a pass, checked on the reads [0, n_slice)
    against the oracle's restatements: diag_filter_v3 lines, and -- with the bad-SUNK flags the device computed over the whole
    set -- the per-read `ID \\t rname` rows of process-by-contig, contig by contig (a read's rows depend on its own hits and on
    the bad set only; the BED is contig-wide and is not compared on a slice)."""
    lr = engine.LocRows.from_sunkset(sunks)
    one_pass(table, reads)
    rows = engine.RowSet(0, 1 << 20)
    rows.append_pass(table, reads, 0, lr)
    n_reads = rs.n_reads
    rlen = (rs.off[1:] - rs.off[:-1]).astype(np.int64)
    raw = rows.download()
    assert rows.n == len(raw["name"]) > 1_000_000 and np.all(np.diff(raw["name"].astype(np.int64)) >= 0)

    def text_of(cols, mask):
        return "".join(f"r{n:07d}\t{p}\t{contig_names[c]}\t{s}\t{g}\n" for n, p, c, s, g in
                       zip(*(cols[x][mask].tolist() for x in ("name", "pos", "contig", "start", "id"))))

    seg_name, bc, bg, bd, kept = rows.diag_filter(contig_names, np.ones(len(contig_names), np.uint8))
    exp = oracle.diag_filter_v3(oracle.parse_sunkpos(text_of(raw, raw["name"] < n_slice)), contig_names)
    sm = seg_name < n_slice
    got = "".join(f"r{n:07d}\t{contig_names[c]}\t{g}\t{chr(d)}\t{g}\n" for n, c, g, d in
                  zip(seg_name[sm].tolist(), bc[sm].tolist(), bg[sm].tolist(), bd[sm].tolist()) if c != _lib.GS_NONE)
    assert got == exp and exp.count("\n") > n_slice // 4
    del raw
    rows.keep_filtered()
    assert rows.n == kept
    f = rows.download()
    bad, meancov, d_bad = engine.RowSet.bad_keys(rows, rows, np.full(lr.n_locs, 3, np.uint8), keep_on_device=True)
    res = rows.pair_stage(np.arange(n_reads, dtype=np.uint32), (rlen >= 10000).astype(np.uint8), d_key_bad=d_bad)
    engine.free_device(d_bad)
    fm = f["name"] < n_slice
    rlen_text = "".join(f"r{i:07d}\t{int(rlen[i])}\n" for i in range(n_slice))
    checked = 0
    for c in np.unique(f["contig"][fm]).tolist():
        m = fm & (f["contig"] == c)
        bm = m & (bad[f["key"]] != 0)
        bad_text = "".join(sorted({f"{contig_names[c]}:{g}\n" for g in f["id"][bm].tolist()}))
        etsv, _ = oracle.process_by_contig(text_of(f, m), rlen_text, bad_text)
        rm = (res["row_contig"] == c) & (res["row_name"] < n_slice)
        tsv = "".join(f"{i}\tr{n:07d}\n" for i, n in zip(res["row_id"][rm].tolist(), res["row_name"][rm].tolist()))
        assert (tsv or f"{contig_names[c]}\n") == etsv, contig_names[c]
        checked += tsv.count("\n")
    assert checked > n_slice // 8
    rows.close()
    lr.close()
    return int(kept), int(len(res["row_id"])), int(len(res["region_start"]))
