"""Drop-in for GAVISUNK's `python workflow/scripts/process-by-contig_lowmem_AR.py <loc> <sunkpos>
<rlen> <bad_sunks> <out.tsv> <out.bed>` (rule process_by_contig, workflow/rules/tagONT.smk:189).

A thin user of the device-resident chain (csrc/gs_chain.hip, gs_rowset_pair_stage): the `.sunkpos` rows are parsed natively
(gs_sunkpos_load), uploaded once, and everything the script does to them runs on the GPU:

  lines 50-108   file-level filtering (bad SUNKs, reads with >= 2 distinct IDs, sort by (rname, start),
                 exact duplicates, reads >= 10 kb)                         -> radix sorts / selections on the device
  lines 135-200  per read: pair test, orientation, multi-position clean-up,
                 largest component                                         -> k_pair_count / k_pair_edges / k_pair_graph
  lines 206-207  `ID \\t rname` TSV
  lines 219-260  contig-wide union of the reads' ID sets, components with > 2 SUNKs -> `contig start end` BED
                                                        -> union-find on the device

The host keeps what is text: the names (their sorted order, `groupby("rname")`), the `.rlen` and bad-SUNK files (native
parsers, gs_sunkpos_names_long / gs_sunkpos_rows_bad) and the two output files.

`--minlen` and `--opt_filt` are accepted and ignored, as in the reference (minlen is overwritten with
10000 at line 106, opt_filt is never read).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import _lib, engine


def stage(sunkpos_path: str, rlen_path: str, bad_path: str, device: int = 0):
    """-> (contig name of the file, names, result dict of RowSet.pair_stage)."""
    L = _lib.lib()
    sp = engine.SunkPos(sunkpos_path)
    if sp.n_rows == 0:
        raise SystemExit("empty sunkpos file")
    contig = sp.contigs[sp.row_contig[0]]  # `sunkposcat['chrom'].unique().tolist()[0]` (:62)
    n_names = L.gs_sunkpos_n_names(sp._h)
    name_rank = engine._view(L.gs_sunkpos_name_rank(sp._h), n_names, np.uint32)
    name_long = np.zeros(max(n_names, 1), np.uint8)
    _lib.check(L.gs_sunkpos_names_long(sp._h, rlen_path.encode(), 10000, name_long.ctypes.data))
    row_bad = np.zeros(max(sp.n_rows, 1), np.uint8)
    _lib.check(L.gs_sunkpos_rows_bad(sp._h, bad_path.encode(), row_bad.ctypes.data))
    # the script groups by read name only (a per-contig file holds one contig): one contig id for every row
    rs = engine.RowSet.from_sunkpos(sp, device, contig=np.zeros(sp.n_rows, np.uint32))
    res = rs.pair_stage(name_rank, name_long[:n_names], row_bad=row_bad[: sp.n_rows])
    rs.close()
    names = [L.gs_sunkpos_name(sp._h, int(i)).decode() for i in np.unique(res["row_name"])]
    name_of = dict(zip(np.unique(res["row_name"]).tolist(), names))
    sp.close()
    return contig, name_of, res


def pair_component(read_off, pos, start, ident, device: int = 0):
    """gs_pair_component over host arrays (rows of read r = [read_off[r], read_off[r+1]), already filtered and sorted):
    -> (IDs kept per read, their values from out_ids[read_off[r]] on)."""
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


def run(loc, sunkpos, rlen, bad, out_tsv, out_bed, device: int = 0) -> int:
    contig, name_of, res = stage(sunkpos, rlen, bad, device)
    if not len(res["row_id"]):  # :92-94, :202-204: a one-line file with the contig name, and no BED
        with open(out_tsv, "w") as fh:
            fh.write(f"{contig}\n")
        return 0
    with open(out_tsv, "w") as fh:
        fh.write("".join(f"{i}\t{name_of[n]}\n" for i, n in zip(res["row_id"].tolist(), res["row_name"].tolist())))
    with open(out_bed, "w") as fh:
        for s, e in zip(res["region_start"].tolist(), res["region_end"].tolist()):
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
