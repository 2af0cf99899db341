"""Drop-in for GAVISUNK's `workflow/scripts/kmerpos_annot3 <reads> <kmers.db> <kmer.loc> <out>`
(rule SUNK_annot, workflow/rules/tagONT.smk:36; source workflow/src/kmerpos_annot3.nim).

Same four positionals, same output file (`name \\t pos \\t chrom \\t start \\t group`), same two
stderr progress lines (kmerpos_annot3.nim:40,72), exit code 1 when a read hits a db k-mer that
has no loc row (the reference's KeyError).  Tagging runs on the GPU through libgavisunk.so.

Environment: GAVISUNK_DEVICE (GPU index, default 0), GAVISUNK_BATCH_BASES (bases per device
batch, default 2^31), GAVISUNK_PARSE=host (records parsed by the native host reader instead of on the GPU; the default
falls back to it by itself for texts outside the regular FASTA / four-line FASTQ forms).

Prebuilt sidecar: the rule runs this program once per scatter chunk (tagONT.smk:19-37) and the reference parses <kmers.db> and
<kmer.loc> every time (kmerpos_annot3.nim:20-72: 1.5e9 text rows each at genome scale).  When `<kmer.loc>.gsidx` exists and
its fingerprints of the two files still hold, the parsed location rows and the BUILT device table are taken from it instead
(gs_sidecar_open); a stale or foreign one is ignored silently.  It is written by `--prepare <kmers.db> <kmer.loc>` (no reads:
parse, build, write) or by any run with GAVISUNK_WRITE_SIDECAR=1.  The output bytes are the same either way.

`--emulate-chunks N` (after the four positionals): <reads> is a WHOLE haplotype's read file
and <out> becomes what the workflow's scatter produces for it -- the reads dealt to N chunk files record by record
(`seqtk seq -F '#' | rustybam fastq-split`, tagONT.smk:2-18; round-robin, as rustybam is believed to deal them: the tool is
not in the build image), the tagger run per chunk with a fresh `prevLoc` (tagONT.smk:19-37), outputs concatenated in
scatter order (tagONT.smk:112-131) -- in ONE run: every device batch is tagged in chunk-major order in a single pass, each
chunk's carried location seeded from what that chunk carried out of the previous batch.
"""
from __future__ import annotations

import os
import queue
import sys
import threading

import time

from . import _lib, engine

# seconds of the last run() by phase (inflate / read of the text, copy to the device, record parse on the device, tagging,
# formatting + writing the rows, waiting for the reader thread); what bench.py's `cli_genome_scale` key reports
last_timing: dict = {}


def _tick(key: str, t0: float) -> float:
    t1 = time.perf_counter()
    last_timing[key] = last_timing.get(key, 0.0) + (t1 - t0)
    return t1


class _Prefetch:
    """Read batches parsed and 2-bit packed by a second thread (the native parser runs without the GIL), one batch ahead:
    the first batch is ready when the k-mer list and the location table are, later ones while the GPU tags their
    predecessors.  Errors of the reads file surface where the reference meets them: when the batch is asked for."""

    def __init__(self, path: str, max_bases: int, text: bool = False, device: int = 0, bgzf_ok: bool = True):
        """text=True: blocks of the file's text, already copied to the device (engine.text_blocks -> engine.StagedText, for the
        on-device record parser) instead of parsed batches; bgzf_ok=False: a `.gz` is read by the host reader whatever its shape"""
        self._q: queue.Queue = queue.Queue(maxsize=1)
        self._stop = threading.Event()
        self._text, self._device = text, device
        self._bgzf_ok = bgzf_ok
        self.used_bgzf = False
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
            bgzf = None
            if self._text and self._bgzf_ok and path.endswith(".gz") and os.environ.get("GAVISUNK_GZ", "device") == "device":
                try:  # BGZF (what rustybam fastq-split / bgzip write): the compressed bytes go to the device and are inflated THERE
                    bgzf = engine.BgzfReader(path, self._device)
                except _lib.GsError as e:
                    if e.status != _lib.GS_E_IRREGULAR:
                        raise
            if bgzf is not None:
                self.used_bgzf = True

                def inflated():
                    # (blocks of up to 3 GiB of text: ~5e4 members, a lane each, fill the device's wave slots in one launch)
                    it = bgzf.blocks(max_bases if max_bases < (1 << 30) else 3 << 30)
                    try:
                        while True:
                            t0 = time.perf_counter()
                            st = next(it, None)
                            _tick("read_and_inflate_on_device", t0)
                            if st is None:
                                return
                            yield st
                    finally:
                        bgzf.close()
                src = inflated()
            elif self._text:
                # two stages of their own: the text of block i + 1 is read / inflated (native threads, no GIL) while block i is
                # copied to the device
                tq: queue.Queue = queue.Queue(maxsize=1)

                def inflate():
                    try:
                        it = engine.text_blocks(path, max_bases)
                        while not self._stop.is_set():
                            t0 = time.perf_counter()
                            t = next(it, None)
                            _tick("read_or_inflate_text", t0)
                            while not self._stop.is_set():
                                try:
                                    tq.put(t, timeout=0.1)
                                    break
                                except queue.Full:
                                    pass
                            if t is None:
                                return
                    except BaseException as e:
                        tq.put(e)

                threading.Thread(target=inflate, daemon=True).start()

                def staged():
                    while True:
                        t = tq.get()
                        if t is None:
                            return
                        if isinstance(t, BaseException):
                            raise t
                        t0 = time.perf_counter()
                        st = engine.StagedText(t, self._device)
                        _tick("copy_text_to_device", t0)
                        yield st
                src = staged()
            else:
                src = engine.read_batches(path, max_bases=max_bases)
            for b in src:
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


class _Rows:
    """Where a batch's rows become text: on the device, from the pass's hit list where it lies (gs_rowwriter_rows_device; the
    location rows go to the device once, when the first batch has rows), or -- GAVISUNK_ROWS=host -- on the host from the hits
    the pass copied back (gs_rowwriter_rows, one thread, three lookups per row into the host arrays of the location table)."""

    def __init__(self, out: "engine.RowWriter", db, table, device: int):
        self.out, self.db, self.table, self.device = out, db, table, device
        self.on_device = os.environ.get("GAVISUNK_ROWS", "device") != "host"
        self.lr = None

    def write(self, hits, dev, names_of=None) -> None:
        if not len(hits):
            return
        if not self.on_device:
            self.out.write(hits, names_of if names_of is not None else dev, self.db)
            return
        if self.lr is None:
            t0 = time.perf_counter()
            self.lr = engine.LocRows.from_db(self.db, self.device)
            _tick("loc_rows_to_device", t0)
        self.out.write_device(self.table, dev, self.lr, self.db.contigs, names_of)

    def close(self) -> None:
        if self.lr is not None:
            self.lr.close()
            self.lr = None


def _tag_text_blocks(blocks, db, table, ofile: str, device: int):
    """The reads file as text blocks parsed on the GPU (gs_reads_from_text).  -> exit code, or None
    when the text is outside the regular FASTA / FASTQ subset the device parser takes (nothing useful is written then)."""
    carry = _lib.GS_NONE
    with engine.RowWriter(ofile) as out:  # (rows formatted and written natively: gs_rowwriter_rows)
        # the rows of block i are formatted and written by a second thread while block i + 1 is parsed and tagged (the hit list of a
        # pass lives in pinned memory of ITS reads handle until that handle is closed: the main thread closes a block's handle only
        # when the writer has handed it back)
        todo: "queue.Queue" = queue.Queue(maxsize=2)
        done: "queue.Queue" = queue.Queue()
        failed = []
        rows = _Rows(out, db, table, device)

        def writer():
            while True:
                item = todo.get()
                if item is None:
                    return
                hits, dev = item
                try:
                    if not failed:
                        t1 = time.perf_counter()
                        rows.write(hits, dev)
                        _tick("write_rows", t1)
                except BaseException as e:  # noqa: BLE001 (handed to the main thread)
                    failed.append(e)
                done.put(dev)

        def close_done() -> None:
            while True:
                try:
                    done.get_nowait().close()
                except queue.Empty:
                    return

        wt = threading.Thread(target=writer, name="gavisunk-rows", daemon=True)
        wt.start()
        rc = 0
        try:
            it = iter(blocks)
            while True:
                t0 = time.perf_counter()
                staged = next(it, None)
                t0 = _tick("wait_for_reader", t0)
                if staged is None:
                    break
                try:
                    dev = engine.DeviceReads.from_staged(staged, db.k)
                except _lib.GsError as e:
                    staged.close()
                    if e.status == _lib.GS_E_IRREGULAR:
                        rc = None
                        break
                    raise
                staged.close()
                t0 = _tick("parse_records_on_device", t0)
                st, hits, carry, _ = engine.tag(table, dev, carry)
                t0 = _tick("tag", t0)
                last_timing["rows"] = last_timing.get("rows", 0) + len(hits)
                todo.put((hits, dev))
                close_done()
                _tick("release_batch", t0)
                if failed:
                    break
                if st == _lib.GS_E_KEYERROR:
                    sys.stderr.write("Error: unhandled exception: key not found [KeyError]\n")
                    rc = 1
                    break
        finally:
            todo.put(None)
            t0 = time.perf_counter()
            wt.join()
            _tick("wait_for_writer", t0)
            close_done()
            rows.close()
        if failed:
            raise failed[0]
    return rc


def _tag_host_batches(batches, db, table, ofile: str, batch_bases: int) -> int:
    """Batches parsed and packed by the host reader.  A file that is one batch is tagged in one call; a longer one goes through
    the streaming engine (gs_stream_*): batch i + 1 is uploaded while batch i is tagged, the rows of batch i are formatted and
    written while batch i + 1 is tagged, and `prevLoc` goes from pass to pass on the device."""
    it = iter(batches)
    first = next(it, None)
    second = next(it, None) if first is not None else None
    with engine.RowWriter(ofile) as out:
        if first is None:
            return 0
        rows = _Rows(out, db, table, table.device)
        if second is None:
            dev = engine.DeviceReads(first, db.k, table.device)
            st, hits, _, _ = engine.tag(table, dev)
            rows.write(hits, dev, names_of=first)
            rows.close()
            dev.close()
            first.close()
            if st == _lib.GS_E_KEYERROR:
                sys.stderr.write("Error: unhandled exception: key not found [KeyError]\n")
                return 1
            return 0
        # the reader's own limits: a batch has at most 2^22 reads and batch_bases bases (+ 2 words per read)
        cap_reads = min(1 << 22, max(batch_bases, 4096))
        cap_words = batch_bases // 32 + 2 * cap_reads + 64
        stream = engine.Stream(table, cap_words, cap_reads, depth=3)
        pending = []  # batches in flight (names are needed when the rows come back), with the device copy of an outsized one

        def drain(upto: int) -> int:
            while len(pending) > upto:
                st, hits, _, _ = stream.next(copy=False)
                b, dev = pending.pop(0)
                out.write(hits, b, db)  # (a stream slot's hit list is not a handle of ours: these rows are formatted on the host)
                if dev is not None:
                    dev.close()
                b.close()
                if st == _lib.GS_E_KEYERROR:
                    sys.stderr.write("Error: unhandled exception: key not found [KeyError]\n")
                    return 1
            return 0

        rc = 0
        try:
            import itertools

            for batch in itertools.chain((first, second), it):
                rc = drain(1)  # at most two in flight: one being tagged, one being uploaded
                if rc:
                    batch.close()
                    break
                if batch.n_words > cap_words or batch.n_reads > cap_reads:  # (one read longer than a batch: its own device copy)
                    dev = engine.DeviceReads(batch, db.k, table.device)
                    stream.submit_reads(dev)
                    pending.append((batch, dev))
                else:
                    stream.submit_batch(batch)
                    pending.append((batch, None))
            if not rc:
                rc = drain(0)
        finally:
            for b, dev in pending:
                if dev is not None:
                    dev.close()
                b.close()
            stream.close()
            rows.close()
        return rc


def run(reads: str, kmers: str, locs: str, ofile: str, device: int = 0, batch_bases: int = 1 << 31) -> int:
    # (a reads "file" that can be read only once -- a FIFO, /dev/stdin -- goes through the host reader: one gzopen, any text shape;
    # the device parser's route hands texts it does not take back to be read AGAIN)
    host_parse = os.environ.get("GAVISUNK_PARSE", "device") == "host" or not engine.is_regular_file(reads)
    last_timing.clear()
    t_run = time.perf_counter()
    # (round 5 tried blocking device waits for .gz inputs -- hipDeviceScheduleBlockingSync, to leave the CPUs to the inflate threads: 3 %
    # on the host reader, and an intermittent deadlock inside the HIP runtime with three threads on the device: profiles/r05/NOTES.md)
    blocking = False
    batches = _Prefetch(reads, batch_bases) if host_parse else None
    blocks = None if host_parse else _Prefetch(reads, min(batch_bases, 1 << 30), text=True, device=device)  # (read and copied meanwhile)
    try:
        pre = engine.open_sidecar(kmers, locs, device)  # the prebuilt rows + table, when valid
        if pre is None:
            db = engine.SunkDB(kmers, locs)
    except _lib.GsError as e:
        for pf in (batches, blocks):
            if pf is not None:
                pf.close()
        sys.stderr.write(e.msg + "\n")
        return 1
    sys.stderr.write("FINISHED READING KMERS\n")
    sys.stderr.write("FINISHED READING LOCS\n")
    t_run = _tick("load_db_and_loc", t_run)
    if pre is not None:
        db, table = pre
    else:
        table = engine.DeviceTable.from_db(db, device)
        _tick("build_table", t_run)
        if os.environ.get("GAVISUNK_WRITE_SIDECAR") == "1":
            try:
                engine.write_sidecar(db, table, kmers, locs)
            except (_lib.GsError, OSError) as e:  # (a read-only directory must not fail the run)
                sys.stderr.write(f"sidecar not written: {e}\n")
    rc = None
    if not host_parse:  # records parsed on the GPU; texts it does not take (multi-line FASTQ, ...) go to the host parser
        try:
            rc = _tag_text_blocks(blocks, db, table, ofile, device)
        except OSError as e:
            sys.stderr.write(f"cannot open the file {reads}: {e}\n")
            rc = 1
        except _lib.GsError as e:
            # the native gzip reader refuses a damaged / truncated .gz (GS_E_IO) or runs out of memory on it (GS_E_NOMEM) where
            # gzip.open raised BadGzipFile, an OSError: the same message and exit code (what was written so far stays, as it
            # does when the reference dies in the middle of a file)
            if e.status == _lib.GS_E_IRREGULAR and blocks.used_bgzf:
                # a file that starts in bgzip's container and goes on as plain gzip members (still a valid gzip file): once more from
                # the start with the host reader (the output file is opened anew: nothing of the first attempt stays)
                blocks.close()
                last_timing.clear()
                blocks = _Prefetch(reads, min(batch_bases, 1 << 30), text=True, device=device, bgzf_ok=False)
                try:
                    rc = _tag_text_blocks(blocks, db, table, ofile, device)
                except OSError as e2:
                    sys.stderr.write(f"cannot open the file {reads}: {e2}\n")
                    rc = 1
                except _lib.GsError as e2:
                    if e2.status not in (_lib.GS_E_IO, _lib.GS_E_NOMEM):
                        raise
                    sys.stderr.write(f"cannot open the file {reads}: {e2.msg}\n")
                    rc = 1
            elif e.status not in (_lib.GS_E_IO, _lib.GS_E_NOMEM):
                raise
            else:
                sys.stderr.write(f"cannot open the file {reads}: {e.msg}\n")
                rc = 1
        blocks.close()
        if rc is None:
            batches = _Prefetch(reads, batch_bases)
    if rc is None:
        rc = _tag_host_batches(batches, db, table, ofile, batch_bases)
    if batches is not None:
        batches.close()
    table.close()
    db.close()
    if blocking:  # (a caller that goes on in this process -- bench.py -- gets its spinning waits back)
        _lib.lib().gs_device_blocking_sync(device, 0)
    return rc


def run_chunked(reads: str, kmers: str, locs: str, ofile: str, nchunks: int, device: int = 0, batch_bases: int = 1 << 31) -> int:
    """`--emulate-chunks N`.  Every device batch is tagged once, in chunk-major order; its rows are formatted on the device as ONE
    text (gs_rowwriter_rows_device_cuts) into a spill file beside <out>, with the stretch of every scatter chunk known; at the end the
    stretches are copied into <out> chunk by chunk (what `combine_ont` does with the per-chunk files, tagONT.smk:112-131).  Nothing
    of a pass is held as Python objects per row or per read."""
    import numpy as np

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
    carry = [_lib.GS_NONE] * nchunks          # per chunk: the location its reads so far carry (prevLoc of that chunk's run)
    pieces = [[] for _ in range(nchunks)]     # per chunk: (offset, bytes) of its stretches in the spill file, in batch order
    # the spill file: beside <out> (the same file system: what the scatter's per-chunk files would take), or in the temporary
    # directory when <out> is no regular file (a FIFO, /dev/stdout) or its directory cannot be written to
    import tempfile

    where = os.path.dirname(os.path.abspath(ofile))
    if not engine.is_regular_file(ofile) or not os.access(where, os.W_OK):
        where = tempfile.gettempdir()
    fd, spill = tempfile.mkstemp(prefix=os.path.basename(ofile) + ".", suffix=".chunks.tmp", dir=where)
    os.close(fd)
    rec = 0                                   # records seen: record i goes to chunk i mod nchunks
    at = 0                                    # bytes in the spill file
    rc = 0
    lr = None
    try:
        with engine.RowWriter(spill) as out:
            for batch in batches:
                n = batch.n_reads
                chunk_of = (rec + np.arange(n, dtype=np.int64)) % nchunks
                order = np.argsort(chunk_of, kind="stable").astype(np.uint32)  # chunk-major, file order inside a chunk
                present = np.unique(chunk_of)
                starts = np.searchsorted(chunk_of[order], present)              # tagging position of each present chunk's first read
                dev = engine.DeviceReads.ordered(batch, order, db.k, device)
                dev.set_seeds(starts, [carry[int(c)] for c in present])
                st, hits, last, _ = engine.tag(table, dev)
                seg = dev.segment_carries(table)  # seg[j] = carried up to starts[j] = what chunk present[j-1] carries now
                for j, c in enumerate(present.tolist()):
                    carry[c] = int(seg[j + 1]) if j + 1 < len(present) else int(last)
                if len(hits):
                    if lr is None:
                        lr = engine.LocRows.from_db(db, device)
                    blob, off, ln = batch.names_blob()
                    bounds = np.searchsorted(hits["read"], np.append(starts, n)).astype(np.uint64)  # rows of every present chunk
                    cuts = out.write_device(table, dev, lr, db.contigs, names=(blob, off[order], ln[order]), row_cuts=bounds)
                    for j, c in enumerate(present.tolist()):
                        if cuts[j + 1] > cuts[j]:
                            pieces[c].append((at + int(cuts[j]), int(cuts[j + 1] - cuts[j])))
                    at += int(cuts[-1])
                dev.close()
                batch.close()
                rec += n
                if st == _lib.GS_E_KEYERROR:
                    sys.stderr.write("Error: unhandled exception: key not found [KeyError]\n")
                    rc = 1
                    break
        with open(ofile, "wb") as dst, open(spill, "rb") as src:
            for c in range(nchunks):
                for off_b, n_b in pieces[c]:
                    while n_b:
                        try:
                            m = os.sendfile(dst.fileno(), src.fileno(), off_b, min(n_b, 1 << 30))
                        except (OSError, AttributeError):
                            m = 0
                        if m <= 0:  # (an output sendfile does not take: through a buffer)
                            src.seek(off_b)
                            buf = src.read(min(n_b, 1 << 24))
                            if not buf:
                                raise OSError("short read of " + spill)
                            dst.write(buf)
                            dst.flush()
                            m = len(buf)
                        off_b += m
                        n_b -= m
    finally:
        try:
            os.unlink(spill)
        except OSError:
            pass
        if lr is not None:
            lr.close()
        batches.close()
        table.close()
        db.close()
    return rc


def prepare(kmers: str, locs: str, device: int = 0) -> int:
    """`--prepare <kmers.db> <kmer.loc>`: parse, build, write the sidecar (what every later run over the two files then skips)."""
    try:
        db = engine.SunkDB(kmers, locs)
        table = engine.DeviceTable.from_db(db, device)
        path = engine.write_sidecar(db, table, kmers, locs)
    except _lib.GsError as e:
        sys.stderr.write(e.msg + "\n")
        return 1
    sys.stderr.write(f"{path}: {len(db.row_contig)} location rows, table {table.layout.split(':')[0]} ({table.nbytes / 2**20:.0f} MiB)\n")
    table.close()
    db.close()
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 3 and argv[0] == "--prepare":
        return prepare(argv[1], argv[2], device=int(os.environ.get("GAVISUNK_DEVICE", "0")))
    nchunks = 0
    if len(argv) == 6 and argv[4] == "--emulate-chunks":
        nchunks = int(argv[5])
        argv = argv[:4]
    if len(argv) != 4 or nchunks < 0:
        sys.stderr.write("usage: kmerpos_annot3 <reads.fa|fq[.gz]> <kmers.db> <kmer.loc> <out.sunkpos> [--emulate-chunks N]\n")
        return 2
    dev, bb = int(os.environ.get("GAVISUNK_DEVICE", "0")), int(os.environ.get("GAVISUNK_BATCH_BASES", str(1 << 31)))
    if nchunks:
        return run_chunked(argv[0], argv[1], argv[2], argv[3], nchunks, device=dev, batch_bases=bb)
    return run(argv[0], argv[1], argv[2], argv[3], device=dev, batch_bases=bb)


if __name__ == "__main__":
    sys.exit(main())
