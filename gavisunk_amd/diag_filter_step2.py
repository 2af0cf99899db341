"""Drop-in for GAVISUNK's `workflow/scripts/diag_filter_step2 <sunkpos> <diag>` > stdout (rule
diag_filter_final, workflow/rules/tagONT.smk:110; source workflow/src/diag_filter_step2.nim):
copies through the `.sunkpos` rows whose contig is the best contig recorded for their read.

A text join with no arithmetic: it stays on the host, in the native library like the reference's compiled program
(gs_diag_filter_step2_files).  (Inside one process the same mask comes out of gs_diag_filter as `keep[]`, without a round
trip through files.)  Both inputs may be gzip."""
from __future__ import annotations

import os
import sys
import tempfile

from . import _lib


def run(sunkpos: str, diag: str, out=None) -> int:
    """out = None: the rows go to the process's stdout (as the rule's `> {output}` expects); a file object: to it."""
    L = _lib.lib()
    if out is None:
        sys.stdout.flush()
        _lib.check(L.gs_diag_filter_step2_files(sunkpos.encode(), diag.encode(), None, None))
        return 0
    fd, tmp = tempfile.mkstemp(prefix="gs_step2_")
    os.close(fd)
    try:
        _lib.check(L.gs_diag_filter_step2_files(sunkpos.encode(), diag.encode(), tmp.encode(), None))
        with open(tmp, "r") as fh:
            out.write(fh.read())
    finally:
        os.unlink(tmp)
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        sys.stderr.write("usage: diag_filter_step2 <reads.sunkpos> <diag.sunkpos>  > <diag2.sunkpos>\n")
        return 2
    return run(argv[0], argv[1])


if __name__ == "__main__":
    sys.exit(main())
