"""Drop-in for GAVISUNK's `python workflow/scripts/badsunks_AR.py <fai1> <fai2> <sunkpos1> <sunkpos2> <out>`
(rule bad_sunks, workflow/rules/tagONT.smk:150).

Per haplotype: reads per `chrom:ID` (GPU histogram, gs_key_histogram), "mean coverage" = mode of the counts
on this haplotype's contigs (smallest mode on ties, as `Series.mode().values[0]`), limit = mode + 4*sqrt(mode);
bad = own-haplotype SUNKs with count > limit or count < 2, other-haplotype SUNKs with count > limit
(badsunks_AR.py:43-51).  Output = union over both haplotypes, one `chrom:ID` per line.  The reference writes
a Python set in arbitrary order; this stub writes the same set sorted."""
from __future__ import annotations

import os
import sys

import numpy as np

from . import _lib, engine


def bad_of_hap(fai: str, sunkpos: str, device: int = 0):
    with open(fai) as fh:
        contigs = {ln.split("\t")[0].rstrip("\n") for ln in fh if ln.strip()}
    sp = engine.SunkPos(sunkpos)
    pairs = np.stack([sp.row_contig.astype(np.int64), sp.row_group], axis=1)
    uniq, inv = np.unique(pairs, axis=0, return_inverse=True)
    inv = np.ascontiguousarray(inv.reshape(-1), np.uint32)
    # `ID2.str.split(":", 1)` (badsunks_AR.py:29): the "contig" that is looked up in the fai is the text before the FIRST
    # colon of `chrom:ID`, so a contig called chr1:1-5000 is looked up as chr1
    sel = np.array([1 if sp.contigs[c].split(":", 1)[0] in contigs else 0 for c in uniq[:, 0].tolist()] + [0], dtype=np.uint8)
    n_keys = len(uniq)
    counts = np.zeros(max(n_keys, 1), np.uint32)
    hist_len = 1 << 16
    hist = np.zeros(hist_len, np.uint32)
    _lib.check(_lib.lib().gs_key_histogram(device, len(inv), inv.ctypes.data if len(inv) else None, n_keys,
                                           sel.ctypes.data, counts.ctypes.data, hist.ctypes.data, hist_len))
    counts = counts[:n_keys]
    if hist[:-1].sum() + hist[-1] == 0:
        raise SystemExit("badsunks: no SUNK of this haplotype's contigs in " + sunkpos)  # the script dies on mode()[0] too
    if hist[-1] and hist[-1] >= hist.max():
        meancov = int(np.bincount(counts[sel[:n_keys] == 1]).argmax())  # absurdly deep data: exact mode on the host
    else:
        meancov = int(hist[:-1].argmax())
    limit = meancov + (meancov ** 0.5) * 4
    names = [f"{sp.contigs[c]}:{i}" for c, i in uniq.tolist()]
    own = sel[:n_keys] == 1
    bad = {names[j] for j in np.nonzero(own & ((counts > limit) | (counts < 2)))[0]}
    bad |= {names[j] for j in np.nonzero(~own & (counts > limit))[0]}
    sp.close()
    return meancov, bad


def run(fai1, fai2, sunkpos1, sunkpos2, outpath, device: int = 0) -> int:
    m1, b1 = bad_of_hap(fai1, sunkpos1, device)
    print("mean coverage", m1)
    m2, b2 = bad_of_hap(fai2, sunkpos2, device)
    print("mean coverage", m2)
    allbad = b1 | b2
    print(len(allbad))
    with open(outpath, "w") as fh:
        for e in sorted(allbad):
            fh.write(e + "\n")
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 5:
        sys.stderr.write("usage: badsunks <hap1.fai> <hap2.fai> <hap1.sunkpos> <hap2.sunkpos> <bad_sunks.txt>\n")
        return 2
    return run(*argv, device=int(os.environ.get("GAVISUNK_DEVICE", "0")))


if __name__ == "__main__":
    sys.exit(main())
