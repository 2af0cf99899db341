"""Drop-in for GAVISUNK's `workflow/scripts/diag_filter_v3 <sunkpos> <fai>` > stdout (rule
diag_filter_step, workflow/rules/tagONT.smk:92; source workflow/src/diag_filter_v3.nim): one line
`rname \\t bestcontig \\t N \\t dir \\t N` per read whose best contig of this haplotype has N > 1
distinct SUNK groups on the median diagonal.  A thin user of the device-resident chain: the rows are parsed natively,
uploaded once, and the whole filter (Table slot order included) runs on the GPU (gs_rowset_diag_filter).  The row set's
columns are uint32 (contig coordinates, as the Python stage reads them: process-by-contig_lowmem_AR.py:60); the Nim program
parses pos / start with parseInt (diag_filter_v3.nim:148-150), so a file with a value outside uint32 -- a negative start --
takes gs_diag_filter instead: the same kernels over 64-bit columns, the Table order replayed on the host."""
from __future__ import annotations

import os
import sys

from . import _lib, engine


def run(sunkpos: str, fai: str, out=None, device: int = 0) -> int:
    out = out or sys.stdout
    L = _lib.lib()
    with open(fai) as fh:
        contigs = {ln.split("\t")[0].rstrip("\n") for ln in fh if ln.strip()}
    sp = engine.SunkPos(sunkpos)
    if not all(len(c) == 0 or (int(c.min()) >= 0 and int(c.max()) <= 0xFFFFFFFF) for c in (sp.row_pos, sp.row_start, sp.row_group)):
        bc, bg, bd, _ = engine.diag_filter(sp, contigs, device)
        out.write("".join(f"{sp.seg_name(seg)}\t{sp.contigs[bc[seg]]}\t{bg[seg]}\t{chr(bd[seg])}\t{bg[seg]}\n"
                          for seg in range(sp.n_segs) if bc[seg] != _lib.GS_NONE))
        sp.close()
        return 0
    rs = engine.RowSet.from_sunkpos(sp, device)
    seg_name, bc, bg, bd, _ = rs.diag_filter(sp.contigs, [1 if c in contigs else 0 for c in sp.contigs])
    rs.close()
    none = _lib.GS_NONE
    out.write("".join(f"{L.gs_sunkpos_name(sp._h, n).decode()}\t{sp.contigs[c]}\t{g}\t{chr(d)}\t{g}\n"
                      for n, c, g, d in zip(seg_name.tolist(), bc.tolist(), bg.tolist(), bd.tolist()) if c != none))
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
