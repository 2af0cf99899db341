"""Drop-in for GAVISUNK's `workflow/scripts/rlen <reads> <out>` (rule read_lengths,
workflow/rules/tagONT.smk:73; source workflow/src/rlen.nim:13-14): `name \\t len(sequence)` per
record.  No GPU involved.

A regular file is read as blocks of whole records (engine.text_blocks: `.gz` through the native parallel gzip reader, all host
threads) and only SCANNED for names and line lengths (gs_rowwriter_rlen: no base is packed or copied).  Text outside the regular
FASTA / four-line FASTQ forms, a damaged `.gz`, a FIFO: the record reader with kseq's rules (gs_fastx_*, one zlib thread, what the
reference does) reads the file from its start instead -- the output is the same either way.  GAVISUNK_RLEN=reader forces it."""
from __future__ import annotations

import os
import sys

import numpy as np

from . import _lib, engine


def _scan(reads: str, ofile: str) -> bool:
    """-> False when the file has to be read by the record reader"""
    try:
        with engine.RowWriter(ofile) as out:
            for blk in engine.text_blocks(reads, 1 << 30):
                out.rlen(np.frombuffer(blk, dtype=np.uint8))
        return True
    except (_lib.GsError, OSError, ValueError):
        return False


def run(reads: str, ofile: str) -> int:
    if os.environ.get("GAVISUNK_RLEN", "scan") != "reader" and engine.is_regular_file(reads) and _scan(reads, ofile):
        return 0
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
