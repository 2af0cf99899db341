"""Drop-in for GAVISUNK's `python workflow/scripts/badsunks_AR.py <fai1> <fai2> <sunkpos1> <sunkpos2> <out>`
(rule bad_sunks, workflow/rules/tagONT.smk:150).  A thin user of the device-resident chain
(gs_rowset_bad_keys): both files' rows are uploaded once, keyed by the dense id of their `chrom:ID`.

Per haplotype: reads per `chrom:ID`, "mean coverage" = mode of the counts on this haplotype's contigs (smallest mode on ties,
as `Series.mode().values[0]`), limit = mode + 4*sqrt(mode); bad = own-haplotype SUNKs with count > limit or count < 2,
other-haplotype SUNKs with count > limit (badsunks_AR.py:43-51).  Output = union over both haplotypes, one `chrom:ID` per
line.  The reference writes a Python set in arbitrary order; this stub writes the same set sorted."""
from __future__ import annotations

import os
import sys

import numpy as np

from . import _lib, engine


def _fai(path):
    with open(path) as fh:
        return {ln.split("\t")[0].rstrip("\n") for ln in fh if ln.strip()}


def run(fai1, fai2, sunkpos1, sunkpos2, outpath, device: int = 0) -> int:
    fais = [_fai(fai1), _fai(fai2)]
    sps = [engine.SunkPos(sunkpos1), engine.SunkPos(sunkpos2)]
    # one key space over both files: the distinct (contig name, ID) pairs
    cname = sorted(set(sps[0].contigs) | set(sps[1].contigs))
    cid = {c: i for i, c in enumerate(cname)}
    parts = []
    for sp in sps:
        key = np.zeros(max(sp.n_rows, 1), np.uint32)
        import ctypes as C
        pc, pi = C.c_void_p(), C.c_void_p()
        nk = _lib.lib().gs_sunkpos_row_keys(sp._h, key.ctypes.data, C.byref(pc), C.byref(pi))
        kc = engine._view(pc.value, nk, np.uint32)
        ki = engine._view(pi.value, nk, np.int64)
        g = np.array([cid[c] for c in sp.contigs] + [0], dtype=np.int64)[kc]  # file-local contig id -> global
        parts.append((key[: sp.n_rows], (g << 40) ^ ki))
    allk = np.unique(np.concatenate([p[1] for p in parts]))
    sets = []
    for sp, (key, gk) in zip(sps, parts):
        rs = engine.RowSet(device, sp.n_rows)
        gkey = np.searchsorted(allk, gk).astype(np.uint32)[key] if sp.n_rows else np.zeros(0, np.uint32)
        z = np.zeros(sp.n_rows, np.uint32)
        rs.append_host(z, z, z, z, z, gkey)  # (only the keys matter here)
        sets.append(rs)
    # `ID2.str.split(":", 1)` (badsunks_AR.py:29): the "contig" that is looked up in the fai is the text before the FIRST colon of
    # `chrom:ID`, so a contig called chr1:1-5000 is looked up as chr1
    kcontig = (allk >> 40).astype(np.int64)
    in1 = np.array([1 if c.split(":", 1)[0] in fais[0] else 0 for c in cname], dtype=np.uint8)[kcontig] if len(allk) else np.zeros(0, np.uint8)
    in2 = np.array([1 if c.split(":", 1)[0] in fais[1] else 0 for c in cname], dtype=np.uint8)[kcontig] if len(allk) else np.zeros(0, np.uint8)
    if not len(allk):
        raise SystemExit("badsunks: no rows in " + sunkpos1 + " / " + sunkpos2)
    try:
        bad, meancov, _ = engine.RowSet.bad_keys(sets[0], sets[1], in1 | (in2 << 1))
    except _lib.GsError as e:
        raise SystemExit("badsunks: " + e.msg)  # (the script dies on mode()[0] too)
    print("mean coverage", meancov[0])
    print("mean coverage", meancov[1])
    w = np.nonzero(bad)[0]
    names = sorted(f"{cname[int(c)]}:{int(i)}" for c, i in zip(kcontig[w].tolist(), (allk[w] & ((1 << 40) - 1)).tolist()))
    print(len(names))
    with open(outpath, "w") as fh:
        fh.write("".join(e + "\n" for e in names))
    for x in sets + sps:
        x.close()
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 5:
        sys.stderr.write("usage: badsunks <hap1.fai> <hap2.fai> <hap1.sunkpos> <hap2.sunkpos> <bad_sunks.txt>\n")
        return 2
    return run(*argv, device=int(os.environ.get("GAVISUNK_DEVICE", "0")))


if __name__ == "__main__":
    sys.exit(main())
