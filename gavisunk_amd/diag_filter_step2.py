"""Drop-in for GAVISUNK's `workflow/scripts/diag_filter_step2 <sunkpos> <diag>` > stdout (rule
diag_filter_final, workflow/rules/tagONT.smk:110; source workflow/src/diag_filter_step2.nim):
copies through the `.sunkpos` rows whose contig is the best contig recorded for their read.

A text join with no arithmetic: it stays on the host.  (Inside one process the same mask comes out
of gs_diag_filter as `keep[]`, without a round trip through files.)  Both inputs may be gzip."""
from __future__ import annotations

import gzip
import sys


def _open(path: str):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "r")


def run(sunkpos: str, diag: str, out=None) -> int:
    out = out or sys.stdout
    best = {}
    with _open(diag) as fh:  # diag_filter_step2.nim:16-22; a later line for the same read overwrites
        for ln in fh:
            f = ln.rstrip("\n").split("\t")
            if len(f) >= 2:
                best[f[0]] = f[1]
    prev, cur = "string", ""  # diag_filter_step2.nim:41-42 starts with rname_prev = "string"
    with _open(sunkpos) as fh:
        for ln in fh:
            ln = ln.rstrip("\n")
            if not ln:
                continue
            f = ln.split("\t")
            if f[0] != prev:
                cur = best.get(f[0], "")
            prev = f[0]
            if len(f) > 2 and f[2] == cur:
                out.write("\t".join(f) + "\n")
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        sys.stderr.write("usage: diag_filter_step2 <reads.sunkpos> <diag.sunkpos>  > <diag2.sunkpos>\n")
        return 2
    return run(argv[0], argv[1])


if __name__ == "__main__":
    sys.exit(main())
