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
import queue
import sys
import threading

from . import _lib, engine


class _Prefetch:
    """Read batches parsed and 2-bit packed by a second thread (the native parser runs without the GIL), one batch ahead:
    the first batch is ready when the k-mer list and the location table are, later ones while the GPU tags their
    predecessors.  Errors of the reads file surface where the reference meets them: when the batch is asked for."""

    def __init__(self, path: str, max_bases: int):
        self._q: queue.Queue = queue.Queue(maxsize=1)
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._work, args=(path, max_bases), daemon=True)
        self._t.start()

    def _put(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def _work(self, path, max_bases):
        try:
            for b in engine.read_batches(path, max_bases=max_bases):
                if not self._put(b):
                    b.close()
                    return
            self._put(None)
        except BaseException as e:  # handed to the consumer
            self._put(e)

    def __iter__(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        self._stop.set()
        while True:  # drop what was parsed ahead
            try:
                item = self._q.get_nowait()
            except queue.Empty:
                break
            if item is not None and not isinstance(item, BaseException):
                item.close()
        self._t.join(timeout=5)


def run(reads: str, kmers: str, locs: str, ofile: str, device: int = 0, batch_bases: int = 1 << 31) -> int:
    batches = _Prefetch(reads, batch_bases)
    try:
        db = engine.SunkDB(kmers, locs)
    except _lib.GsError as e:
        batches.close()
        sys.stderr.write(e.msg + "\n")
        return 1
    sys.stderr.write("FINISHED READING KMERS\n")
    sys.stderr.write("FINISHED READING LOCS\n")
    table = engine.DeviceTable.from_db(db, device)
    carry = _lib.GS_NONE
    rc = 0
    with open(ofile, "w") as out:
        for batch in batches:
            dev = engine.DeviceReads(batch, db.k, device)
            st, hits, carry, _ = engine.tag(table, dev, carry)
            out.write(engine.format_sunkpos(hits, batch.names, db))
            dev.close()
            batch.close()
            if st == _lib.GS_E_KEYERROR:
                sys.stderr.write("Error: unhandled exception: key not found [KeyError]\n")
                rc = 1
                break
    batches.close()
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
