"""Drop-in for GAVISUNK's `workflow/scripts/kmerpos_annot3 <reads> <kmers.db> <kmer.loc> <out>`
(rule SUNK_annot, workflow/rules/tagONT.smk:36; source workflow/src/kmerpos_annot3.nim).

Same four positionals, same output file (`name \\t pos \\t chrom \\t start \\t group`), same two
stderr progress lines (kmerpos_annot3.nim:40,72), exit code 1 when a read hits a db k-mer that
has no loc row (the reference's KeyError).  Tagging runs on the GPU through libgavisunk.so.

Environment: GAVISUNK_DEVICE (GPU index, default 0), GAVISUNK_BATCH_BASES (bases per device
batch, default 2^31).
"""
from __future__ import annotations

import os
import sys

from . import _lib, engine


def run(reads: str, kmers: str, locs: str, ofile: str, device: int = 0, batch_bases: int = 1 << 31) -> int:
    try:
        db = engine.SunkDB(kmers, locs)
    except _lib.GsError as e:
        sys.stderr.write(e.msg + "\n")
        return 1
    sys.stderr.write("FINISHED READING KMERS\n")
    sys.stderr.write("FINISHED READING LOCS\n")
    table = engine.DeviceTable.from_db(db, device)
    carry = _lib.GS_NONE
    rc = 0
    with open(ofile, "w") as out:
        for batch in engine.read_batches(reads, max_bases=batch_bases):
            dev = engine.DeviceReads(batch, db.k, device)
            st, hits, carry, _ = engine.tag(table, dev, carry)
            out.write(engine.format_sunkpos(hits, batch.names, db))
            dev.close()
            batch.close()
            if st == _lib.GS_E_KEYERROR:
                sys.stderr.write("Error: unhandled exception: key not found [KeyError]\n")
                rc = 1
                break
    table.close()
    db.close()
    return rc


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 4:
        sys.stderr.write("usage: kmerpos_annot3 <reads.fa|fq[.gz]> <kmers.db> <kmer.loc> <out.sunkpos>\n")
        return 2
    return run(argv[0], argv[1], argv[2], argv[3], device=int(os.environ.get("GAVISUNK_DEVICE", "0")),
               batch_bases=int(os.environ.get("GAVISUNK_BATCH_BASES", str(1 << 31))))


if __name__ == "__main__":
    sys.exit(main())
