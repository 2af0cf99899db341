"""Drop-in for GAVISUNK's `workflow/scripts/diag_filter_v3 <sunkpos> <fai>` > stdout (rule
diag_filter_step, workflow/rules/tagONT.smk:92; source workflow/src/diag_filter_v3.nim): one line
`rname \\t bestcontig \\t N \\t dir \\t N` per read whose best contig of this haplotype has N > 1
distinct SUNK groups on the median diagonal.  The statistics run on the GPU (gs_diag_filter)."""
from __future__ import annotations

import os
import sys

from . import _lib, engine


def run(sunkpos: str, fai: str, out=None, device: int = 0) -> int:
    out = out or sys.stdout
    with open(fai) as fh:
        contigs = [ln.split("\t")[0].rstrip("\n") for ln in fh if ln.strip()]
    sp = engine.SunkPos(sunkpos)
    bc, bg, bd, _ = engine.diag_filter(sp, contigs, device)
    for seg in range(sp.n_segs):
        if bc[seg] != _lib.GS_NONE:
            out.write(f"{sp.seg_name(seg)}\t{sp.contigs[bc[seg]]}\t{bg[seg]}\t{chr(bd[seg])}\t{bg[seg]}\n")
    sp.close()
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        sys.stderr.write("usage: diag_filter_v3 <reads.sunkpos> <hap.fai>  > <diag.sunkpos>\n")
        return 2
    return run(argv[0], argv[1], device=int(os.environ.get("GAVISUNK_DEVICE", "0")))


if __name__ == "__main__":
    sys.exit(main())
