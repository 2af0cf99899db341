"""Drop-in for GAVISUNK's `python workflow/scripts/process-by-contig_lowmem_AR.py <loc> <sunkpos>
<rlen> <bad_sunks> <out.tsv> <out.bed>` (rule process_by_contig, workflow/rules/tagONT.smk:189).

  lines 50-108   file-level filtering (bad SUNKs, reads with >= 2 distinct IDs, sort by (rname, start),
                 exact duplicates, reads >= 10 kb)                         -> host, NumPy (it is pandas there)
  lines 135-200  per read: pair test, orientation, multi-position clean-up,
                 largest component                                         -> GPU, gs_pair_component
  lines 206-207  `ID \\t rname` TSV
  lines 219-260  contig-wide union of the reads' ID sets, components with > 2 SUNKs -> `contig start end` BED
                                                        -> GPU, gs_union_regions

`--minlen` and `--opt_filt` are accepted and ignored, as in the reference (minlen is overwritten with
10000 at line 106, opt_filt is never read).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import _lib, engine


def load_rlen(path: str) -> dict:
    out = {}
    with open(path) as fh:
        for ln in fh:
            f = ln.rstrip("\n").split("\t")
            if len(f) >= 2:
                out[f[0]] = int(f[1])
    return out


def prepare(sunkpos_path: str, rlen_path: str, bad_path: str):
    """Lines 50-108 -> (contig, names[sorted], read_off, pos, start, id) or (contig, None...) when empty."""
    sp = engine.SunkPos(sunkpos_path)
    if sp.n_rows == 0:
        raise SystemExit("empty sunkpos file")
    contig = sp.contigs[sp.row_contig[0]]
    with open(bad_path) as fh:
        bad = set(fh.read().splitlines())
    seg_names = [sp.seg_name(i) for i in range(sp.n_segs)]
    uniq = {}
    seg_uid = np.array([uniq.setdefault(n, len(uniq)) for n in seg_names], dtype=np.int64)
    names = list(uniq)
    rid = seg_uid[sp.row_seg]
    pos = sp.row_pos.astype(np.int64)
    start = sp.row_start.astype(np.int64)
    ident = sp.row_group.astype(np.int64)
    chrom = sp.row_contig.astype(np.int64)
    # ID2 = chrom:ID not in bad (:66-72)
    if bad:
        badset = set()
        for b in bad:
            c, _, i = b.rpartition(":")
            badset.add((c, i))
        keep = np.array([(sp.contigs[c], str(i)) not in badset for c, i in zip(chrom.tolist(), ident.tolist())], dtype=bool)
    else:
        keep = np.ones(len(pos), dtype=bool)
    rid, pos, start, ident, chrom = rid[keep], pos[keep], start[keep], ident[keep], chrom[keep]
    sp.close()
    # reads with at least two distinct IDs (:91-97)
    multi = np.zeros(len(names), dtype=bool)
    if len(rid):
        pairs = np.unique(np.stack([rid, ident], axis=1), axis=0)
        cnt = np.bincount(pairs[:, 0], minlength=len(names))
        multi = cnt > 1
    if not multi.any():
        return contig, None
    m = multi[rid]
    rid, pos, start, ident, chrom = rid[m], pos[m], start[m], ident[m], chrom[m]
    # sort by (rname, start), stable (:100); rname order = string order
    order_of_name = np.empty(len(names), dtype=np.int64)
    order_of_name[np.argsort(np.array(names, dtype=object), kind="stable")] = np.arange(len(names))
    key = order_of_name[rid]
    o = np.lexsort((start, key))
    rid, pos, start, ident, chrom, key = rid[o], pos[o], start[o], ident[o], chrom[o], key[o]
    # exact duplicate rows, keep first (:104)
    rows = np.stack([key, pos, chrom, start, ident], axis=1)
    _, first = np.unique(rows, axis=0, return_index=True)
    sel = np.sort(first)
    rid, pos, start, ident, key = rid[sel], pos[sel], start[sel], ident[sel], key[sel]
    # reads of at least 10 kb (:106-108)
    rl = load_rlen(rlen_path)
    long_enough = np.array([rl.get(n, -1) >= 10000 for n in names], dtype=bool)
    m = long_enough[rid]
    rid, pos, start, ident, key = rid[m], pos[m], start[m], ident[m], key[m]
    if not len(rid):
        return contig, ([], np.zeros(1, np.uint64), pos, start, ident)
    # groups in sorted-name order (:135)
    change = np.concatenate(([True], key[1:] != key[:-1]))
    starts_idx = np.nonzero(change)[0]
    read_off = np.concatenate((starts_idx, [len(rid)])).astype(np.uint64)
    gnames = [names[rid[i]] for i in starts_idx]
    return contig, (gnames, read_off, pos, start, ident)


def pair_component(read_off, pos, start, ident, device: int = 0):
    n_reads = len(read_off) - 1
    out_n = np.zeros(max(n_reads, 1), np.uint32)
    out_ids = np.zeros(max(len(pos), 1), np.int64)
    pos = np.ascontiguousarray(pos, np.int64)
    start = np.ascontiguousarray(start, np.int64)
    ident = np.ascontiguousarray(ident, np.int64)
    read_off = np.ascontiguousarray(read_off, np.uint64)
    _lib.check(_lib.lib().gs_pair_component(device, n_reads, read_off.ctypes.data, pos.ctypes.data, start.ctypes.data,
                                            ident.ctypes.data, out_n.ctypes.data, out_ids.ctypes.data))
    return out_n[:n_reads], out_ids


def contig_regions(per_read_ids, device: int = 0):
    """Lines 219-260: union of the reads' ID sets (GPU union-find, gs_union_regions); components with more than
    two SUNKs -> (start, end), sorted by start."""
    ids = np.array([v for r in per_read_ids for v in r], dtype=np.int64)
    if not len(ids):
        return []
    row_read = np.repeat(np.arange(len(per_read_ids), dtype=np.uint32), [len(r) for r in per_read_ids])
    vertex_id, row_vertex = np.unique(ids, return_inverse=True)
    row_vertex = np.ascontiguousarray(row_vertex.reshape(-1), np.uint32)
    row_read = np.ascontiguousarray(row_read, np.uint32)
    vertex_id = np.ascontiguousarray(vertex_id, np.int64)
    nv = len(vertex_id)
    size = np.zeros(nv, np.uint32)
    cmin = np.zeros(nv, np.int64)
    cmax = np.zeros(nv, np.int64)
    _lib.check(_lib.lib().gs_union_regions(device, len(ids), row_read.ctypes.data, row_vertex.ctypes.data, nv,
                                           vertex_id.ctypes.data, size.ctypes.data, cmin.ctypes.data, cmax.ctypes.data))
    keep = np.nonzero(size > 2)[0]
    return sorted((int(cmin[i]), int(cmax[i])) for i in keep)


def run(loc, sunkpos, rlen, bad, out_tsv, out_bed, device: int = 0) -> int:
    contig, prep = prepare(sunkpos, rlen, bad)
    per_read = []
    if prep is not None and len(prep[0]):
        gnames, read_off, pos, start, ident = prep
        out_n, out_ids = pair_component(read_off, pos, start, ident, device)
        for r, nm in enumerate(gnames):
            k = int(out_n[r])
            if k:
                lo = int(read_off[r])
                per_read.append((nm, out_ids[lo:lo + k].tolist()))
    if not per_read:  # :92-94, :202-204: a one-line file with the contig name, and no BED
        with open(out_tsv, "w") as fh:
            fh.write(f"{contig}\n")
        return 0
    with open(out_tsv, "w") as fh:
        for nm, ids in per_read:
            fh.write("".join(f"{i}\t{nm}\n" for i in ids))
    with open(out_bed, "w") as fh:
        for s, e in contig_regions([ids for _, ids in per_read], device):
            fh.write(f"{contig}\t{s}\t{e}\n")
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("SUNKs")
    ap.add_argument("sunkpos")
    ap.add_argument("rlen")
    ap.add_argument("badsunks")
    ap.add_argument("outputfile")
    ap.add_argument("outputbed")
    ap.add_argument("--minlen", default=10000, type=int)
    ap.add_argument("--opt_filt", action="store_true")
    a = ap.parse_args(argv)
    return run(a.SUNKs, a.sunkpos, a.rlen, a.badsunks, a.outputfile, a.outputbed,
               device=int(os.environ.get("GAVISUNK_DEVICE", "0")))


if __name__ == "__main__":
    sys.exit(main())
