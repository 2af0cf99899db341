"""Drop-in for GAVISUNK's `workflow/scripts/rlen <reads> <out>` (rule read_lengths,
workflow/rules/tagONT.smk:73; source workflow/src/rlen.nim:13-14): `name \\t len(sequence)` per
record.  A by-product of the native read parser; no GPU involved."""
from __future__ import annotations

import sys

from . import engine


def run(reads: str, ofile: str) -> int:
    with open(ofile, "w") as out:
        for batch in engine.read_batches(reads, max_bases=1 << 28):
            out.write("".join(f"{n}\t{l}\n" for n, l in zip(batch.names, batch.len.tolist())))
            batch.close()
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        sys.stderr.write("usage: rlen <reads.fa|fq[.gz]> <out.rlen>\n")
        return 2
    return run(argv[0], argv[1])


if __name__ == "__main__":
    sys.exit(main())
