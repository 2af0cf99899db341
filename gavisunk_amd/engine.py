"""Host-side mirror of the reference's tagging programs on top of the C-ABI.

`SunkDB`      = what kmerpos_annot3.nim:20-69 loads (SUNK list + loc table)
`ReadBatch`   = records of `readfq(reads)` (kmerpos_annot3.nim:85), 2-bit packed
`Tagger`      = the main loop kmerpos_annot3.nim:82-96 on one GPU, batch after batch with the
                `prevLoc` carry threaded through
All compute happens in libgavisunk.so's HIP kernels; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterator, List, Sequence

import numpy as np

from . import _lib
from ._lib import GS_NONE, GsCounters, GsHit, check, lib

HIT_DTYPE = np.dtype([("read", "<u4"), ("pos", "<u4"), ("payload", "<u4")])
assert HIT_DTYPE.itemsize == C.sizeof(GsHit)


def _view(ptr: int, n: int, dtype) -> np.ndarray:
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class SunkDB:
    """Parsed `<kmers.db>` + `<kmer.loc>` (native parser, gs_sunkdb_load)."""

    def __init__(self, db_path: str, loc_path: str):
        h = C.c_void_p()
        check(lib().gs_sunkdb_load(db_path.encode(), loc_path.encode(), C.byref(h)))
        self._h = h
        L = lib()
        self.k = L.gs_sunkdb_k(h)
        nr = L.gs_sunkdb_n_rows(h)
        self.row_contig = _view(L.gs_sunkdb_row_contig(h), nr, np.uint32)
        self.row_start = _view(L.gs_sunkdb_row_start(h), nr, np.int64)
        self.row_group = _view(L.gs_sunkdb_row_group(h), nr, np.int64)
        self.row_loc = _view(L.gs_sunkdb_row_loc(h), nr, np.uint32)
        self.contigs = [L.gs_sunkdb_contig_name(h, i).decode() for i in range(L.gs_sunkdb_n_contigs(h))]

    # the join of the k-mer list with the location table -- distinct k-mers, each with its winning loc row -- is made on the
    # device for the table (DeviceTable.from_db: gs_table_create_from_db); these three make it on the host, on first use
    @property
    def keys(self) -> np.ndarray:
        L = lib()
        return _view(L.gs_sunkdb_keys(self._h), L.gs_sunkdb_n_keys(self._h), np.uint64)

    @property
    def key_row(self) -> np.ndarray:
        L = lib()
        return _view(L.gs_sunkdb_key_row(self._h), L.gs_sunkdb_n_keys(self._h), np.uint32)

    @property
    def key_loc(self) -> np.ndarray:
        L = lib()
        return _view(L.gs_sunkdb_key_loc(self._h), L.gs_sunkdb_n_keys(self._h), np.uint32)

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_sunkdb_free(self._h)
            self._h = None

    __del__ = close

    @classmethod
    def _wrap(cls, h) -> "SunkDB":
        self = cls.__new__(cls)
        self._h = h
        L = lib()
        self.k = L.gs_sunkdb_k(h)
        nr = L.gs_sunkdb_n_rows(h)
        self.row_contig = _view(L.gs_sunkdb_row_contig(h), nr, np.uint32)
        self.row_start = _view(L.gs_sunkdb_row_start(h), nr, np.int64)
        self.row_group = _view(L.gs_sunkdb_row_group(h), nr, np.int64)
        self.row_loc = _view(L.gs_sunkdb_row_loc(h), 0, np.uint32)
        self.contigs = [L.gs_sunkdb_contig_name(h, i).decode() for i in range(L.gs_sunkdb_n_contigs(h))]
        return self


def sidecar_path(loc_path: str) -> str:
    return loc_path + ".gsidx"


def open_sidecar(db_path: str, loc_path: str, device: int = 0):
    """(SunkDB, DeviceTable) from the prebuilt sidecar of the two files (gs_sidecar_open), or None when there is none or it is
    stale: the caller then parses and builds as the reference does on every run (kmerpos_annot3.nim:20-72)."""
    side = sidecar_path(loc_path)
    if not os.path.exists(side):
        return None
    dbh, th = C.c_void_p(), C.c_void_p()
    st = lib().gs_sidecar_open(side.encode(), db_path.encode(), loc_path.encode(), device, C.byref(dbh), C.byref(th))
    if st == _lib.GS_E_IO:
        return None
    check(st)
    db = SunkDB._wrap(dbh)
    t = DeviceTable.__new__(DeviceTable)
    t._h, t.k, t.device, t.n = th, db.k, device, None
    return db, t


def write_sidecar(db: "SunkDB", table: "DeviceTable", db_path: str, loc_path: str) -> str:
    side = sidecar_path(loc_path)
    check(lib().gs_sidecar_write(side.encode(), db._h, table._h, db_path.encode(), loc_path.encode()))
    return side


class ReadBatch:
    """Host-side packed batch (gs_batch)."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        L = lib()
        self.n_reads = L.gs_batch_n_reads(handle)
        self.n_bases = L.gs_batch_n_bases(handle)
        self.n_words = L.gs_batch_n_words(handle)
        self.words = _view(L.gs_batch_words(handle), self.n_words, np.uint64)
        self.word_off = _view(L.gs_batch_word_off(handle), self.n_reads + 1, np.uint64)
        self.len = _view(L.gs_batch_len(handle), self.n_reads, np.uint32)

    @property
    def names(self) -> List[str]:
        L = lib()
        return [L.gs_batch_name(self._h, i).decode() for i in range(self.n_reads)]

    def names_blob(self):
        """-> (bytes of all names as a uint8 view, start of every name uint64[n_reads], length of every name uint32[n_reads]):
        what RowWriter.write_device takes as `names`, without a Python string per read (gs_batch_names)"""
        base, off, nb = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        check(lib().gs_batch_names(self._h, C.byref(base), C.byref(off), C.byref(nb)))
        if not self.n_reads or not nb.value:
            return np.zeros(1, np.uint8), np.zeros(self.n_reads, np.uint64), np.zeros(self.n_reads, np.uint32)
        blob = _view(base.value, nb.value, np.uint8)
        o = _view(off.value, self.n_reads, np.uint64)
        ends = np.append(o[1:], np.uint64(nb.value))
        return blob, o, (ends - o - np.uint64(1)).astype(np.uint32)  # (every name is followed by its NUL)

    @classmethod
    def from_ascii(cls, seqs: Sequence[bytes]) -> "ReadBatch":
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if len(seqs):
            off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        buf = np.frombuffer(b"".join(seqs), dtype=np.uint8) if len(seqs) else np.zeros(1, np.uint8)
        return cls.from_concat(buf, off)

    @classmethod
    def from_concat(cls, buf: np.ndarray, off: np.ndarray) -> "ReadBatch":
        buf = np.ascontiguousarray(buf, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        h = C.c_void_p()
        check(lib().gs_batch_from_ascii(buf.ctypes.data, off.ctypes.data, len(off) - 1, C.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_batch_free(self._h)
            self._h = None

    __del__ = close


def read_batches(path: str, max_bases: int = 1 << 30, max_reads: int = 1 << 22) -> Iterator[ReadBatch]:
    """Iterates a FASTA/FASTQ(.gz) file in packed batches (native parser, kseq semantics)."""
    fx = C.c_void_p()
    check(lib().gs_fastx_open(path.encode(), C.byref(fx)))
    try:
        while True:
            b = C.c_void_p()
            check(lib().gs_fastx_next_batch(fx, max_bases, max_reads, C.byref(b)))
            if not b:
                break
            yield ReadBatch(b)
    finally:
        lib().gs_fastx_close(fx)


class _NativeBuffer:
    """Bytes the native library handed over (gs_gz_read): freed when the last view of them goes."""

    def __init__(self, ptr: int, n: int):
        self.ptr, self.n = ptr, n

    def view(self, lo: int, hi: int):
        arr = (C.c_ubyte * max(hi - lo, 0)).from_address(self.ptr + lo)
        arr._gs_owner = self  # (the array -- and every memoryview / numpy array made from it -- keeps the buffer alive)
        return memoryview(arr).cast("B")

    def __del__(self):
        if self.ptr:
            lib().gs_gz_free_buffer(self.ptr)
            self.ptr = 0


def gz_text_blocks(path: str, max_bytes: int = 1 << 30, threads: int = 0):
    """The inflated text of a gzip-compressed FASTA / FASTQ file in blocks of whole records of about max_bytes: inflated by the
    native parallel reader (csrc/gs_inflate.cpp: several threads, block starts found in the compressed stream, CRC checked),
    cut where gs_text_record_cut says a record ends.  Large blocks are views of the reader's buffers (no copy); small ones are
    bytes.  Raises GsError(GS_E_IO) where zlib would fail too."""
    L = lib()
    z = C.c_void_p()
    check(L.gs_gz_open(path.encode(), threads, C.byref(z)))

    def piece(buf, lo, hi):
        v = buf.view(lo, hi)
        return v if hi - lo >= (1 << 24) else bytes(v)

    try:
        rem = b""
        want = min(max_bytes, 64 << 20)  # the first blocks are small (copy and parser start early), then they double up to max_bytes
        while True:
            p, n = C.c_void_p(), C.c_uint64(0)
            check(L.gs_gz_read(z, want - len(rem) if want > len(rem) else want, rem, len(rem), C.byref(p), C.byref(n)))
            want = min(2 * want, max_bytes)
            buf, n, at_end = _NativeBuffer(p.value or 0, n.value), n.value, bool(L.gs_gz_at_end(z))
            if n == len(rem) and at_end:  # end of file: what is left is the last block
                if n:
                    yield piece(buf, 0, n)
                return
            off = 0
            nl = L.gs_gz_newlines(z) + rem.count(b"\n")  # (counted by the reader while it wrote the bytes: no pass over the text here)
            if n <= 2 * max_bytes:  # the usual case: one block, cut where the last whole record ends
                cut = n if at_end else L.gs_text_record_cut_counted(buf.ptr, n, nl)
                if cut:
                    yield piece(buf, 0, cut)
                rem = bytes(buf.view(cut, n))
                if at_end:
                    if rem:
                        yield rem
                    return
                continue
            while n - off > max_bytes:  # (a read that delivered far more than a block's worth)
                w, cut = max_bytes, 0
                while True:
                    cut = L.gs_text_record_cut(buf.ptr + off, w)
                    if cut or w >= n - off:
                        break
                    w = min(2 * w, n - off)  # a record longer than a block: grow
                if not cut:
                    break
                yield piece(buf, off, off + cut)
                off += cut
            cut = n - off if at_end else L.gs_text_record_cut(buf.ptr + off, n - off)
            if cut:
                yield piece(buf, off, off + cut)
            rem = bytes(buf.view(off + cut, n))
            if at_end and not rem:
                return
            if at_end:  # (text after the last record end of a file that is over: the last block)
                yield rem
                return
    finally:
        L.gs_gz_close(z)


def is_regular_file(path: str) -> bool:
    """False for what can be read only once (a FIFO, a character device, /dev/stdin into a pipe); True when the path does not
    exist either: opening it then fails where it always did."""
    import stat

    try:
        return stat.S_ISREG(os.stat(path).st_mode)
    except OSError:
        return True


def text_blocks(path: str, max_bytes: int = 1 << 30) -> Iterator[bytes]:
    """The (decompressed) text of a FASTA / FASTQ file in blocks of whole records, for DeviceReads.from_text: FASTA blocks end
    before a line that starts with '>', FASTQ blocks after a multiple of four lines (the regular four-line form; anything
    else is refused by gs_reads_from_text and read by the host parser instead).  `.gz` files are inflated by the native
    parallel reader (gz_text_blocks); GAVISUNK_GZ=python keeps Python's gzip module (one zlib thread) instead."""
    import gzip

    import mmap

    if not is_regular_file(path):
        # a FIFO, /dev/stdin, a process substitution: opened ONCE and read to its end, as the reference's gzopen does (a second
        # open would wait for a writer that has gone); gzip through Python's zlib on the one stream
        f = open(path, "rb", buffering=1 << 20)
        if f.peek(2)[:2] == b"\x1f\x8b":
            f = gzip.GzipFile(fileobj=f)
            f.myfileobj = f.fileobj  # (closed with it)
    else:
        with open(path, "rb") as probe:
            gz = probe.read(2) == b"\x1f\x8b"
            size = os.fstat(probe.fileno()).st_size
            if not gz and 0 < size <= max_bytes:  # one block: the file's pages as they are (no copy on the host at all)
                yield mmap.mmap(probe.fileno(), 0, access=mmap.ACCESS_READ)  # (unmapped when its last user lets go of it)
                return
        if gz and os.environ.get("GAVISUNK_GZ", "native") != "python":
            yield from gz_text_blocks(path, max_bytes)
            return
        f = gzip.open(path, "rb") if gz else open(path, "rb")
    try:
        rem = b""
        while True:
            chunk = f.read(max_bytes - len(rem) if max_bytes > len(rem) else max_bytes)  # (a record longer than a block: grow)
            buf = rem + chunk if rem else chunk
            if not chunk:  # end of file: what is left is the last block
                if buf:
                    yield buf
                return
            if buf[:1] == b">":
                cut = buf.rfind(b"\n>") + 1  # 0 when the block holds no second record yet
            else:
                nl = buf.count(b"\n")
                cut = len(buf)
                for _ in range(nl % 4 + 1):  # back to the newline that ends the last whole four-line record
                    cut = buf.rfind(b"\n", 0, cut)
                cut = cut + 1 if nl >= 4 else 0
            if cut <= 0:
                rem = buf
                continue
            yield buf[:cut]
            rem = buf[cut:]
    finally:
        f.close()


class StagedText:
    """The text of a FASTA / FASTQ block copied to the device (gs_text_stage) -- done by the reader thread of the command line
    stub while the main thread still loads the SUNK table; DeviceReads.from_staged parses it there."""

    def __init__(self, text, device: int = 0):
        self.text, self.device = text, device
        buf = np.frombuffer(text, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().gs_text_stage(device, buf.ctypes.data if len(buf) else None, len(buf), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_text_stage_free(self._h)
            self._h = None

    __del__ = close


class DeviceText:
    """A FASTA / FASTQ text block that exists only in device memory (inflated there: BgzfReader); DeviceReads.from_staged parses it."""

    text = None  # (no host copy: DeviceReads.from_staged fetches the read names from the device)

    def __init__(self, h, device: int):
        self._h, self.device = h, device

    def host_copy(self) -> bytes:
        n = C.c_uint64(0)
        check(lib().gs_textstage_copy(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint8)
        check(lib().gs_textstage_copy(self._h, out.ctypes.data, len(out), C.byref(n)))
        return out[: n.value].tobytes()

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_text_stage_free(self._h)
            self._h = None

    __del__ = close


class BgzfReader:
    """DEFLATE on the GPU for BGZF-shaped `.fq.gz` (gs_bgzf_*, csrc/gs_bgzf.hip): blocks of whole records, inflated on the device
    from the compressed bytes.  Raises GsError(GS_E_IRREGULAR) for a file that is not BGZF (gz_text_blocks reads those)."""

    def __init__(self, path: str, device: int = 0):
        h = C.c_void_p()
        check(lib().gs_bgzf_open(path.encode(), device, C.byref(h)))
        self._h, self.device = h, device

    def blocks(self, max_bytes: int = 1 << 30) -> Iterator[DeviceText]:
        L = lib()
        # (a launch inflates 64 MiB of compressed bytes -- ~0.4 GB of text -- or fifty times as much in ~30 ms: blocks of less make no
        # sense.  The first block is small so that the parser and the tagger start early; the next ones grow eightfold up to
        # max_bytes -- every block costs the consumer a few synchronisations: 0.73-0.80 s against 0.84-0.93 s with doubling, on the
        # command line's 20 GB of text; GAVISUNK_BGZF_GROW for experiments)
        want = min(max_bytes, 256 << 20)
        grow = want
        while True:
            st = C.c_void_p()
            check(L.gs_bgzf_next(self._h, grow, C.byref(st)))
            if st.value:
                yield DeviceText(st, self.device)
                want = min(int(os.environ.get("GAVISUNK_BGZF_GROW", "8")) * want, max_bytes)
                grow = want
            elif L.gs_bgzf_at_end(self._h):
                return
            else:  # (not one whole record fit: a read longer than the block)
                if grow >= (3 << 30):
                    raise _lib.GsError(_lib.GS_E_IRREGULAR, "a record of more than 3 GiB of text")
                grow *= 4

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_bgzf_close(self._h)
            self._h = None

    __del__ = close


class DeviceTable:
    """SUNK hash table resident on one GPU (gs_table)."""

    def __init__(self, k: int, keys: np.ndarray, loc: np.ndarray, payload: np.ndarray, device: int = 0):
        keys = np.ascontiguousarray(keys, np.uint64)
        loc = np.ascontiguousarray(loc, np.uint32)
        payload = np.ascontiguousarray(payload, np.uint32)
        assert len(keys) == len(loc) == len(payload)
        h = C.c_void_p()
        check(lib().gs_table_create(device, k, len(keys), keys.ctypes.data, loc.ctypes.data, payload.ctypes.data,
                                    C.byref(h)))
        self._h, self.k, self.device, self.n = h, k, device, len(keys)

    @classmethod
    def from_db(cls, db: SunkDB, device: int = 0) -> "DeviceTable":
        """gs_table_create_from_db: the k-mer list joined with the location table on the device, the table built from that."""
        h = C.c_void_p()
        check(lib().gs_table_create_from_db(device, db._h, C.byref(h)))
        t = cls.__new__(cls)
        t._h, t.k, t.device = h, db.k, device
        t.n = None  # (the number of distinct k-mers is the join's business: gs_sunkdb_n_keys makes it on the host)
        return t

    def clone(self) -> "DeviceTable":
        """A second handle of the same table on its device (gs_table_clone: what a receiving rank of gs_table_broadcast does,
        with device-to-device copies)."""
        h = C.c_void_p()
        check(lib().gs_table_clone(self._h, C.byref(h)))
        t = DeviceTable.__new__(DeviceTable)
        t._h, t.k, t.device, t.n = h, self.k, self.device, self.n
        return t

    @property
    def nbytes(self) -> int:
        return lib().gs_table_bytes(self._h)

    @property
    def layout(self) -> str:
        return lib().gs_table_layout(self._h).decode()

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_table_destroy(self._h)
            self._h = None

    __del__ = close


class DeviceReads:
    """Packed reads resident on one GPU (gs_reads)."""

    def __init__(self, batch: ReadBatch, k: int, device: int = 0):
        h = C.c_void_p()
        check(lib().gs_reads_upload(device, batch.words.ctypes.data if batch.n_words else None, batch.n_words,
                                    batch.word_off.ctypes.data, batch.len.ctypes.data if batch.n_reads else None,
                                    batch.n_reads, k, C.byref(h)))
        self._h, self.k, self.device = h, k, device
        self.n_reads, self.n_bases = batch.n_reads, batch.n_bases
        self.n_windows = lib().gs_reads_n_windows(h)

    @classmethod
    def from_ascii(cls, seqs: Sequence[bytes], k: int, device: int = 0) -> "DeviceReads":
        """Upload unpacked reads and pack them on the GPU (gs_reads_upload_ascii)."""
        off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if len(seqs):
            off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        buf = np.frombuffer(b"".join(seqs), dtype=np.uint8) if len(seqs) and off[-1] else np.zeros(1, np.uint8)
        h = C.c_void_p()
        check(lib().gs_reads_upload_ascii(device, buf.ctypes.data, off.ctypes.data, len(seqs), k, C.byref(h)))
        self = cls.__new__(cls)
        self._h, self.k, self.device = h, k, device
        self.n_reads, self.n_bases = len(seqs), int(off[-1])
        self.n_windows = lib().gs_reads_n_windows(h)
        return self

    @classmethod
    def ordered(cls, batch: ReadBatch, order: np.ndarray, k: int, device: int = 0) -> "DeviceReads":
        """The batch's reads tagged in `order` (gs_reads_upload_ordered): hit.read is then the position in `order`."""
        order = np.ascontiguousarray(order, np.uint32)
        h = C.c_void_p()
        check(lib().gs_reads_upload_ordered(device, batch.words.ctypes.data if batch.n_words else None, batch.n_words,
                                            batch.word_off.ctypes.data, batch.len.ctypes.data if batch.n_reads else None,
                                            batch.n_reads, order.ctypes.data if len(order) else None, len(order), k, C.byref(h)))
        self = cls.__new__(cls)
        self._h, self.k, self.device = h, k, device
        self.n_reads, self.n_bases = len(order), int(batch.len[order].sum()) if len(order) else 0
        self.n_windows = lib().gs_reads_n_windows(h)
        return self

    @classmethod
    def from_staged(cls, staged: "StagedText", k: int) -> "DeviceReads":
        """Parse a text that StagedText has already copied to the device (gs_reads_from_staged_text)."""
        L = lib()
        h, ix = C.c_void_p(), C.c_void_p()
        check(L.gs_reads_from_staged_text(staged._h, k, C.byref(h), C.byref(ix)))
        self = cls._with_index(h, ix, staged.text, k, staged.device)
        if staged.text is None and self.n_reads:
            # the text exists only on the device (BgzfReader): the read names are gathered there and come over as one blob
            off = np.zeros(self.n_reads + 1, np.uint64)
            off[1:] = np.cumsum(self._name_len, dtype=np.uint64)
            blob = np.zeros(max(int(off[-1]), 1), np.uint8)
            check(L.gs_textstage_names(staged._h, self._name_off.ctypes.data, self._name_len.ctypes.data, self.n_reads, blob.ctypes.data, off.ctypes.data))
            self._text, self._name_off = blob, off[:-1].copy()
        return self

    @classmethod
    def _with_index(cls, h, ix, text, k: int, device: int) -> "DeviceReads":
        L = lib()
        self = cls.__new__(cls)
        self._h, self.k, self.device = h, k, device
        n = L.gs_textindex_n_reads(ix)
        # (the names stay where they are in the text: a genome-scale batch has 1e6 of them, and the row writer takes them from there)
        self._name_off = _view(L.gs_textindex_name_off(ix), n, np.uint64).copy()
        self._name_len = _view(L.gs_textindex_name_len(ix), n, np.uint32).copy()
        self._text = text
        self._names = None
        self.len = _view(L.gs_textindex_len(ix), n, np.uint32).copy()
        L.gs_textindex_free(ix)
        self.n_reads, self.n_bases = n, int(self.len.sum())
        self.n_windows = L.gs_reads_n_windows(h)
        return self

    @property
    def names(self) -> List[str]:
        """Read names, file order (made on first use)."""
        if getattr(self, "_names", None) is None:
            text = self._text
            self._names = [bytes(text[o:o + l]).decode() for o, l in zip(self._name_off.tolist(), self._name_len.tolist())]
        return self._names

    @classmethod
    def from_text(cls, text: bytes, k: int, device: int = 0) -> "DeviceReads":
        """Records parsed, gathered and packed on the GPU from the text of a FASTA / FASTQ block (gs_reads_from_text).  Raises
        GsError with code GS_E_IRREGULAR for texts outside the regular subset: read those with read_batches.  The result
        also has `names` and `len` (per read, file order), like a ReadBatch."""
        L = lib()
        h, ix = C.c_void_p(), C.c_void_p()
        buf = np.frombuffer(text, dtype=np.uint8)
        check(L.gs_reads_from_text(device, buf.ctypes.data if len(buf) else None, len(buf), k, C.byref(h), C.byref(ix)))
        return cls._with_index(h, ix, text, k, device)

    def set_seeds(self, seed_at, seed) -> None:
        """Force the carried location to seed[j] before the read at tagging position seed_at[j] (gs_reads_set_seeds)."""
        seed_at = np.ascontiguousarray(seed_at, np.uint32)
        seed = np.ascontiguousarray(seed, np.uint32)
        check(lib().gs_reads_set_seeds(self._h, seed_at.ctypes.data if len(seed_at) else None,
                                       seed.ctypes.data if len(seed) else None, len(seed_at)))
        self._n_seeds = len(seed_at)

    def segment_carries(self, table: "DeviceTable") -> np.ndarray:
        out = np.zeros(max(getattr(self, "_n_seeds", 0), 1), np.uint32)
        check(lib().gs_tag_segment_carries(table._h, self._h, out.ctypes.data))
        return out[: getattr(self, "_n_seeds", 0)]

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_reads_destroy(self._h)
            self._h = None

    __del__ = close


def tag(table: DeviceTable, reads: DeviceReads, carry_in: int = GS_NONE):
    """One tagging pass -> (status, hits[HIT_DTYPE], carry_out, counters dict).
    status is GS_OK or GS_E_KEYERROR (then hits = those emitted before the offending window)."""
    L = lib()
    check(L.gs_tag_launch(table._h, reads._h, carry_in))
    n = C.c_uint64(0)
    carry = C.c_uint32(GS_NONE)
    ctr = GsCounters()
    st = L.gs_tag_finish(table._h, reads._h, None, 0, C.byref(n), C.byref(carry), C.byref(ctr))
    if st not in (_lib.GS_OK, _lib.GS_E_KEYERROR):
        check(st)
    out = np.zeros(n.value, dtype=HIT_DTYPE)
    if n.value:
        st2 = L.gs_tag_hits_to_host(table._h, reads._h, out.ctypes.data, n.value, None)
        if st2 not in (_lib.GS_OK, _lib.GS_E_KEYERROR):
            check(st2)
    return st, out, carry.value, ctr.as_dict()


class PinnedWords:
    """Pinned host memory for packed read words (gs_host_alloc): what Stream.submit hands to the DMA engine without a copy."""

    def __init__(self, n_words: int):
        p = C.c_void_p()
        check(lib().gs_host_alloc(max(n_words, 1) * 8, C.byref(p)))
        self._p, self.n_words = p, n_words
        self.words = _view(p.value, n_words, np.uint64)

    def close(self):
        if getattr(self, "_p", None):
            self.words = None
            lib().gs_host_free(self._p)
            self._p = None

    __del__ = close


class Stream:
    """Streaming engine (gs_stream_*): the main loop kmerpos_annot3.nim:85-96 as a pipeline of read batches against one table.
    `submit` queues a host batch (upload overlapped with the previous batch's pass, `prevLoc` handed on in device memory),
    `next` returns the oldest batch's rows; at most `depth` batches are between the two."""

    def __init__(self, table: DeviceTable, max_words: int, max_reads: int, depth: int = 3, carry_in: int = GS_NONE):
        h = C.c_void_p()
        check(lib().gs_stream_create(table._h, max_words, max_reads, depth, carry_in, C.byref(h)))
        self._h, self.table, self.depth = h, table, depth
        self._cap_words, self._cap_reads = max_words, max_reads
        self._keep = []  # host arrays of batches in flight (the DMA engine reads pinned words where they lie)

    def submit(self, words: np.ndarray, word_off: np.ndarray, length: np.ndarray) -> None:
        """A host batch in the ReadBatch layout (words u64, word_off u64[n + 1], len u32[n])."""
        n = len(length)
        check(lib().gs_stream_submit(self._h, words.ctypes.data if len(words) else None, len(words), word_off.ctypes.data,
                                     length.ctypes.data if n else None, n))
        self._keep.append((words, word_off, length))

    def submit_batch(self, batch: ReadBatch) -> None:
        self.submit(batch.words, batch.word_off, batch.len)
        self._keep[-1] = batch

    def submit_reads(self, reads: DeviceReads) -> None:
        """A batch that already is device-resident."""
        check(lib().gs_stream_submit_reads(self._h, reads._h))
        self._keep.append(reads)

    @property
    def in_flight(self) -> int:
        return lib().gs_stream_in_flight(self._h)

    @property
    def carry(self) -> int:
        return lib().gs_stream_carry(self._h)

    def next(self, copy: bool = True):
        """-> (status, hits, n_reads, counters) of the oldest batch in flight; hits is a view of pinned memory that the next
        call invalidates unless copy=True.  status is GS_OK or GS_E_KEYERROR (hits = the rows before the offending window)."""
        hp, n, nr, ctr = C.c_void_p(), C.c_uint64(0), C.c_uint32(0), GsCounters()
        st = lib().gs_stream_next(self._h, C.byref(hp), C.byref(n), C.byref(nr), C.byref(ctr))
        if st not in (_lib.GS_OK, _lib.GS_E_KEYERROR):
            check(st)
        self._keep.pop(0)
        hits = _view(hp.value, n.value, HIT_DTYPE)
        if copy:  # the rows are ours now: the slot may take the next batch at once
            hits = hits.copy()
            lib().gs_stream_release(self._h)
        return st, hits, nr.value, ctr.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_stream_destroy(self._h)
            self._h = None
            self._keep = []

    __del__ = close


def format_sunkpos(hits: np.ndarray, names: Sequence[str], db: SunkDB) -> str:
    """Rows of the reference's output file: name, i, chrom, start, group (kmerpos_annot3.nim:93)."""
    rows = hits["payload"]
    contig = db.row_contig[rows]
    start = db.row_start[rows]
    group = db.row_group[rows]
    cn = db.contigs
    return "".join(f"{names[r]}\t{p}\t{cn[c]}\t{s}\t{g}\n"
                   for r, p, c, s, g in zip(hits["read"].tolist(), hits["pos"].tolist(), contig.tolist(),
                                            start.tolist(), group.tolist()))


class RowWriter:
    """The tagger's output file, written natively (gs_rowwriter_*): `name \\t i \\t chrom \\t start \\t group` per emitted hit."""

    def __init__(self, path: str):
        h = C.c_void_p()
        check(lib().gs_rowwriter_open(path.encode(), C.byref(h)))
        self._h = h

    def write(self, hits: np.ndarray, reads, db: "SunkDB") -> None:
        """hits of one batch; reads: the DeviceReads (parsed on the device: names taken from its text block) or the ReadBatch
        (host parser) they index into."""
        if not len(hits):
            return
        hits = np.ascontiguousarray(hits)
        if getattr(reads, "_name_off", None) is not None:
            base = np.frombuffer(reads._text, dtype=np.uint8)
            check(lib().gs_rowwriter_rows(self._h, hits.ctypes.data, len(hits), db._h, base.ctypes.data, reads._name_off.ctypes.data,
                                          reads._name_len.ctypes.data, None))
        else:
            check(lib().gs_rowwriter_rows(self._h, hits.ctypes.data, len(hits), db._h, None, None, None, reads._h))

    def write_device(self, table: "DeviceTable", reads, locrows: "LocRows", contig_names: Sequence[str], names_of=None, names=None, row_cuts=None):
        """The rows of the LAST pass over `reads` (DeviceReads), formatted on the device from its hit list where it lies
        (gs_rowwriter_rows_device).  Read names: from the text block a device-parsed batch came from; names_of: the ReadBatch the
        reads came from when they were parsed on the host; names: (bytes as uint8 array, offsets uint64, lengths uint32) otherwise.
        row_cuts (ascending row numbers of the pass): -> the bytes of this call's text in front of each of them (uint64 array)."""
        arr = (C.c_char_p * max(len(contig_names), 1))(*[c.encode() for c in contig_names])
        if names is None and getattr(reads, "_name_off", None) is not None:
            names = (np.frombuffer(reads._text, dtype=np.uint8), reads._name_off, reads._name_len)
        cuts = np.ascontiguousarray(row_cuts if row_cuts is not None else [], np.uint64)
        out = np.zeros(len(cuts), np.uint64)
        pc, po = (cuts.ctypes.data, out.ctypes.data) if len(cuts) else (None, None)
        if names is not None:
            base, off, ln = (np.ascontiguousarray(names[0], np.uint8), np.ascontiguousarray(names[1], np.uint64), np.ascontiguousarray(names[2], np.uint32))
            check(lib().gs_rowwriter_rows_device_cuts(self._h, table._h, reads._h, locrows._h, len(contig_names), arr, base.ctypes.data,
                                                      off.ctypes.data, ln.ctypes.data, len(ln), None, pc, len(cuts), po))
        else:
            check(lib().gs_rowwriter_rows_device_cuts(self._h, table._h, reads._h, locrows._h, len(contig_names), arr, None, None, None, 0,
                                                      names_of._h, pc, len(cuts), po))
        return out if row_cuts is not None else None

    def rlen(self, text: np.ndarray) -> int:
        """`name \\t len` for every record of a block of whole records in a regular form (gs_rowwriter_rlen); GsError with
        GS_E_IRREGULAR otherwise.  -> records written"""
        text = np.ascontiguousarray(text, np.uint8)
        n = C.c_uint64(0)
        check(lib().gs_rowwriter_rlen(self._h, text.ctypes.data if len(text) else None, len(text), C.byref(n)))
        return n.value

    def close(self) -> int:
        n = C.c_uint64(0)
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            check(lib().gs_rowwriter_close(h, C.byref(n)))
        return n.value

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    __del__ = close


class SunkPos:
    """Parsed `.sunkpos` file (native parser, gs_sunkpos_load)."""

    def __init__(self, path: str):
        h = C.c_void_p()
        check(lib().gs_sunkpos_load(path.encode(), C.byref(h)))
        self._h = h
        L = lib()
        self.n_rows, self.n_segs = L.gs_sunkpos_n_rows(h), L.gs_sunkpos_n_segs(h)
        self.row_seg = _view(L.gs_sunkpos_row_seg(h), self.n_rows, np.uint32)
        self.row_pos = _view(L.gs_sunkpos_row_pos(h), self.n_rows, np.int64)
        self.row_contig = _view(L.gs_sunkpos_row_contig(h), self.n_rows, np.uint32)
        self.row_start = _view(L.gs_sunkpos_row_start(h), self.n_rows, np.int64)
        self.row_group = _view(L.gs_sunkpos_row_group(h), self.n_rows, np.int64)
        self.contigs = [L.gs_sunkpos_contig_name(h, i).decode() for i in range(L.gs_sunkpos_n_contigs(h))]

    def seg_name(self, seg: int) -> str:
        return lib().gs_sunkpos_seg_name(self._h, seg).decode()

    def row_text(self, row: int) -> bytes:
        n = C.c_uint32(0)
        p = lib().gs_sunkpos_row_text(self._h, row, C.byref(n))
        return C.string_at(p, n.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_sunkpos_free(self._h)
            self._h = None

    __del__ = close


def diag_filter(sp: SunkPos, fai_contigs, device: int = 0):
    """gs_diag_filter -> (best_contig[seg], best_good[seg], best_dir[seg], keep[row])."""
    fai = set(fai_contigs)
    in_fai = np.array([1 if c in fai else 0 for c in sp.contigs] + [0], dtype=np.uint8)
    bc = np.zeros(max(sp.n_segs, 1), np.uint32)
    bg = np.zeros(max(sp.n_segs, 1), np.uint32)
    bd = np.zeros(max(sp.n_segs, 1), np.uint8)
    keep = np.zeros(max(sp.n_rows, 1), np.uint8)
    check(lib().gs_diag_filter(device, sp._h, in_fai.ctypes.data, bc.ctypes.data, bg.ctypes.data, bd.ctypes.data,
                               keep.ctypes.data))
    return bc[: sp.n_segs], bg[: sp.n_segs], bd[: sp.n_segs], keep[: sp.n_rows]


# ---------------------------------------------------------------------------------------------------
# The device-resident second half (csrc/gs_chain.hip): rows -> diagonal filter -> bad SUNKs -> pair stage
# ---------------------------------------------------------------------------------------------------
class LocRows:
    """kmer.loc rows in device memory (gs_locrows): payload of a hit -> chrom / start / ID / dense `chrom:ID` key."""

    def __init__(self, h, device: int, keepalive=None):
        self._h, self.device, self._keep = h, device, keepalive
        self.n_locs = lib().gs_locrows_n_locs(h)

    @classmethod
    def from_db(cls, db: "SunkDB", device: int = 0) -> "LocRows":
        h = C.c_void_p()
        rc_, rs_, rg_ = (np.ascontiguousarray(db.row_contig, np.uint32), np.ascontiguousarray(db.row_start, np.int64), np.ascontiguousarray(db.row_group, np.int64))
        # (a SunkDB from a sidecar carries no dense (chrom, ID) ids: the rows then serve the output writer, which does not use them)
        rl_ = np.ascontiguousarray(db.row_loc, np.uint32) if len(db.row_loc) == len(rc_) else np.full(len(rc_), _lib.GS_MISSING, np.uint32)
        check(lib().gs_locrows_create(device, len(rc_), rc_.ctypes.data, rs_.ctypes.data, rg_.ctypes.data, rl_.ctypes.data, C.byref(h)))
        return cls(h, device)

    @classmethod
    def from_sunkset(cls, sunks: "DeviceSunkSet") -> "LocRows":
        h = C.c_void_p()
        check(lib().gs_locrows_from_sunkset_dev(sunks.device, sunks._h, C.byref(h)))
        return cls(h, sunks.device)  # (a packed copy: the handle does not refer to the set afterwards)

    def key_contig(self) -> np.ndarray:
        out = np.zeros(max(self.n_locs, 1), np.uint32)
        check(lib().gs_locrows_key_contig(self._h, out.ctypes.data))
        return out[: self.n_locs]

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_locrows_free(self._h)
            self._h = None

    __del__ = close


class RowSet:
    """`.sunkpos` rows as arrays in device memory (gs_rowset), from tagging passes (device to device) or from a parsed file."""

    def __init__(self, device: int = 0, cap_hint: int = 0):
        h = C.c_void_p()
        check(lib().gs_rowset_create(device, cap_hint, C.byref(h)))
        self._h, self.device = h, device

    @property
    def n(self) -> int:
        return lib().gs_rowset_n(self._h)

    def append_pass(self, table: "DeviceTable", reads: "DeviceReads", name_base: int, locrows: LocRows) -> None:
        check(lib().gs_rowset_append_pass(self._h, table._h, reads._h, name_base, locrows._h))

    def append_host(self, name, pos, contig, start, ident, key=None) -> None:
        cols = [np.ascontiguousarray(a, np.uint32) for a in (name, pos, contig, start, ident)]
        k = np.ascontiguousarray(key, np.uint32) if key is not None else None
        check(lib().gs_rowset_append_host(self._h, len(cols[0]), *[c.ctypes.data for c in cols], k.ctypes.data if k is not None else None))

    @classmethod
    def from_sunkpos(cls, sp: "SunkPos", device: int = 0, keys: bool = False, contig=None):
        """The rows of a parsed `.sunkpos` file; name ids = the file's distinct names in first-seen order.  keys=True also numbers
        the distinct (chrom, ID) pairs -> (row set, row_key, key_contig, key_id) (keys are uploaded by the caller: append order).
        contig: a replacement for the contig column (process-by-contig groups by read name only)."""
        L = lib()
        n_names = L.gs_sunkpos_n_names(sp._h)
        row_name = _view(L.gs_sunkpos_row_name(sp._h), sp.n_rows, np.uint32)
        rs = cls(device, sp.n_rows)
        kc = ki = None
        key = None
        if keys:
            key = np.zeros(max(sp.n_rows, 1), np.uint32)
            pc, pi = C.c_void_p(), C.c_void_p()
            nk = L.gs_sunkpos_row_keys(sp._h, key.ctypes.data, C.byref(pc), C.byref(pi))
            kc, ki = _view(pc.value, nk, np.uint32).copy(), _view(pi.value, nk, np.int64).copy()
            key = key[: sp.n_rows]
        for col in (sp.row_pos, sp.row_start, sp.row_group):  # (uint32 like the script's dtypes, process-by-contig_lowmem_AR.py:60)
            if len(col) and (int(col.min()) < 0 or int(col.max()) > 0xFFFFFFFF):
                raise ValueError("a .sunkpos column does not fit uint32 (pos / start / ID are contig coordinates)")
        rs.append_host(row_name, sp.row_pos.astype(np.uint32), sp.row_contig if contig is None else contig, sp.row_start.astype(np.uint32),
                       sp.row_group.astype(np.uint32), key)
        rs.n_names = n_names
        return (rs, kc, ki) if keys else rs

    def download(self):
        n = self.n
        cols = [np.zeros(max(n, 1), np.uint32) for _ in range(6)]
        check(lib().gs_rowset_download(self._h, *[c.ctypes.data for c in cols]))
        return dict(zip(("name", "pos", "contig", "start", "id", "key"), (c[:n] for c in cols)))

    def n_segs(self) -> int:
        ns = C.c_uint32(0)
        check(lib().gs_rowset_segments(self._h, C.byref(ns)))
        return ns.value

    def diag_filter(self, contig_names: Sequence[str], in_fai, cap_in: int = 64, want_cap: bool = False):
        """diag_filter_v3 over the rows, on the device -> (seg_name, best_contig, best_good, best_dir, rows kept).
        cap_in / want_cap: one shard of a read-sharded run (gs_rowset_diag_filter_from): the Table capacity the reads before this
        shard leave behind, and (sixth value) the one this shard leaves."""
        ns = self.n_segs()
        arr = (C.c_char_p * max(len(contig_names), 1))(*[c.encode() for c in contig_names])
        in_fai = np.ascontiguousarray(np.concatenate([np.asarray(in_fai, np.uint8), [0]]), np.uint8)
        sn, bc, bg = (np.zeros(max(ns, 1), np.uint32) for _ in range(3))
        bd = np.zeros(max(ns, 1), np.uint8)
        kept, cap_out = C.c_uint64(0), C.c_uint32(0)
        check(lib().gs_rowset_diag_filter_from(self._h, len(contig_names), arr, in_fai.ctypes.data, cap_in, sn.ctypes.data, bc.ctypes.data,
                                               bg.ctypes.data, bd.ctypes.data, C.byref(kept), C.byref(cap_out)))
        if want_cap:
            return sn[:ns], bc[:ns], bg[:ns], bd[:ns], kept.value, cap_out.value
        return sn[:ns], bc[:ns], bg[:ns], bd[:ns], kept.value

    def keep_flags(self) -> np.ndarray:
        out = np.zeros(max(self.n, 1), np.uint8)
        check(lib().gs_rowset_keep_flags(self._h, out.ctypes.data))
        return out[: self.n]

    def keep_filtered(self) -> None:
        check(lib().gs_rowset_keep_filtered(self._h))

    @staticmethod
    def bad_keys(hap1: "RowSet", hap2: "RowSet", key_hap: np.ndarray, keep_on_device: bool = False, reduce=None):
        """badsunks_AR.py over both haplotypes' filtered rows -> (bad[n_keys] uint8, [meancov1, meancov2], device pointer or None).
        reduce: (function pointer, context) of a gs_reduce_u32_fn -- the reads-per-SUNK counters summed over the ranks of a
        read-sharded run (dist.Comm.reducer())."""
        key_hap = np.ascontiguousarray(key_hap, np.uint8)
        bad = np.zeros(max(len(key_hap), 1), np.uint8)
        mc = (C.c_uint32 * 2)()
        d = C.c_void_p()
        fn, ctx = reduce if reduce is not None else (None, None)
        check(lib().gs_rowset_bad_keys_reduced(hap1._h, hap2._h, len(key_hap), key_hap.ctypes.data, fn, ctx, bad.ctypes.data, mc,
                                               C.byref(d) if keep_on_device else None))
        return bad[: len(key_hap)], [int(mc[0]), int(mc[1])], (d.value if keep_on_device else None)

    def pair_stage(self, name_rank, name_long, d_key_bad=None, row_bad=None):
        """process-by-contig over the rows -> dict(row_contig, row_name, row_id, region_contig, region_start, region_end, contigs_in, counters)."""
        L = lib()
        name_rank = np.ascontiguousarray(name_rank, np.uint32)
        name_long = np.ascontiguousarray(name_long, np.uint8)
        rb = np.ascontiguousarray(row_bad, np.uint8) if row_bad is not None else None
        h = C.c_void_p()
        check(L.gs_rowset_pair_stage(self._h, d_key_bad, rb.ctypes.data if rb is not None else None, len(name_rank),
                                     name_rank.ctypes.data if len(name_rank) else None, name_long.ctypes.data if len(name_long) else None, C.byref(h)))
        return _pairstage_result(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_rowset_free(self._h)
            self._h = None

    __del__ = close


def _pairstage_result(h):
    L = lib()
    try:
        t, nr, nc = L.gs_pairstage_n_rows(h), L.gs_pairstage_n_regions(h), L.gs_pairstage_n_contigs_in(h)
        out = {"row_contig": _view(L.gs_pairstage_row_contig(h), t, np.uint32).copy(), "row_name": _view(L.gs_pairstage_row_name(h), t, np.uint32).copy(),
               "row_id": _view(L.gs_pairstage_row_id(h), t, np.int64).copy(),
               "region_contig": _view(L.gs_pairstage_region_contig(h), nr, np.uint32).copy(),
               "region_start": _view(L.gs_pairstage_region_start(h), nr, np.int64).copy(),
               "region_end": _view(L.gs_pairstage_region_end(h), nr, np.int64).copy(),
               "contigs_in": _view(L.gs_pairstage_contigs_in(h), nc, np.uint32).copy(),
               "counters": _view(L.gs_pairstage_counters(h), 8, np.uint64).copy()}
    finally:
        L.gs_pairstage_free(h)
    return out


def pairstage_from_rows(row_contig, row_name, row_id, contigs_in, device: int = 0):
    """gs_pairstage_from_rows: the root of a read-sharded run merges the TSV rows of all ranks (sorted by contig, then by the reads'
    name order) -> the dict RowSet.pair_stage gives on one GPU: rows as handed in, intervals from the union over all of them."""
    rc = np.ascontiguousarray(row_contig, np.uint32)
    rn = np.ascontiguousarray(row_name, np.uint32)
    ri = np.ascontiguousarray(row_id, np.int64)
    ci = np.ascontiguousarray(contigs_in, np.uint32)
    h = C.c_void_p()
    check(lib().gs_pairstage_from_rows(device, len(rc), rc.ctypes.data if len(rc) else None, rn.ctypes.data if len(rc) else None,
                                       ri.ctypes.data if len(rc) else None, len(ci), ci.ctypes.data if len(ci) else None, C.byref(h)))
    return _pairstage_result(h)


def hits_view(table: "DeviceTable", reads: "DeviceReads") -> np.ndarray:
    """The rows of the last pass over `reads` as a view of the batch's pinned host mirror (gs_tag_hits_view; the next pass over the
    batch overwrites it)."""
    p, n = C.c_void_p(), C.c_uint64(0)
    check(lib().gs_tag_hits_view(table._h, reads._h, C.byref(p), C.byref(n)))
    return _view(p.value, n.value, HIT_DTYPE)


def hits_on_device(table: "DeviceTable", reads: "DeviceReads") -> int:
    """Rows of the last pass over `reads` (its hit list stays in device memory until the batch's next pass): what a RowSet that
    takes them should be sized for."""
    p, n = C.c_void_p(), C.c_uint64(0)
    check(lib().gs_tag_hits_device(table._h, reads._h, C.byref(p), C.byref(n)))
    return n.value


def free_device(ptr, device: int = 0) -> None:
    if ptr:
        check(lib().gs_device_free(device, ptr))


def define_sunks(contigs, k: int, device: int = 0):
    """GPU SUNK definition (gs_sunks_define).
    contigs: [(name, ascii bytes)] of both haplotypes -> synth.SunkSet-like arrays in kmer.loc order."""
    from .synth import SunkSet

    off = np.zeros(len(contigs) + 1, dtype=np.uint64)
    if contigs:
        off[1:] = np.cumsum([len(s) for _, s in contigs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(s for _, s in contigs), dtype=np.uint8) if contigs else np.zeros(1, np.uint8)
    h = C.c_void_p()
    check(lib().gs_sunks_define(device, k, len(contigs), buf.ctypes.data, off.ctypes.data, C.byref(h)))
    L = lib()
    n = L.gs_sunkset_n(h)
    out = SunkSet(k, [nm for nm, _ in contigs],
                  _view(L.gs_sunkset_contig(h), n, np.uint32).copy(), _view(L.gs_sunkset_start(h), n, np.int64).copy(),
                  _view(L.gs_sunkset_code(h), n, np.uint64).copy(), _view(L.gs_sunkset_group(h), n, np.int64).copy())
    L.gs_sunkset_free(h)
    return out


# ---------------------------------------------------------------------------------------------------
# Device-resident inputs (genome-scale runs: nothing crosses PCIe between SUNK definition and tagging).
# The caller holds device memory (e.g. torch tensors) and passes raw device pointers through the C-ABI.
# ---------------------------------------------------------------------------------------------------
class DeviceSunkSet:
    """gs_sunks_define_device: the SUNKs of an assembly whose bases (ASCII) are in device memory; the result stays there."""

    def __init__(self, d_seq_ptr: int, off: np.ndarray, k: int, device: int = 0):
        off = np.ascontiguousarray(off, np.uint64)
        h = C.c_void_p()
        check(lib().gs_sunks_define_device(device, k, len(off) - 1, d_seq_ptr, off.ctypes.data, C.byref(h)))
        self._h, self.k, self.device = h, k, device
        L = lib()
        self.n, self.n_groups = L.gs_sunkset_dev_n(h), L.gs_sunkset_dev_n_groups(h)

    def _host(self, ptr: int, dtype, lo: int = 0, hi: int | None = None) -> np.ndarray:
        hi = self.n if hi is None else hi
        out = np.empty(max(hi - lo, 0), dtype=dtype)
        if len(out):
            check(lib().gs_copy_to_host(self.device, out.ctypes.data, ptr + lo * out.itemsize, out.nbytes))
        return out

    def code(self, lo=0, hi=None):
        return self._host(lib().gs_sunkset_dev_code(self._h), np.uint64, lo, hi)

    def pos(self, lo=0, hi=None):
        return self._host(lib().gs_sunkset_dev_pos(self._h), np.uint64, lo, hi)

    def group(self, lo=0, hi=None):
        return self._host(lib().gs_sunkset_dev_group(self._h), np.uint32, lo, hi)

    def loc(self, lo=0, hi=None):
        return self._host(lib().gs_sunkset_dev_loc(self._h), np.uint32, lo, hi)

    def write_files(self, contig_names: Sequence[str], db_path: str, loc_path: str) -> None:
        """The tagger's two input files (<db>: k-mer list, <loc>: location table), written natively (gs_sunkset_dev_write_files)."""
        arr = (C.c_char_p * len(contig_names))(*[n.encode() for n in contig_names])
        check(lib().gs_sunkset_dev_write_files(self._h, arr, len(contig_names), db_path.encode(), loc_path.encode()))

    def table(self) -> "DeviceTable":
        """The tagger's table straight from the device arrays (payload of a hit = its kmer.loc row)."""
        L = lib()
        h = C.c_void_p()
        check(L.gs_table_create_device(self.device, self.k, self.n, L.gs_sunkset_dev_code(self._h),
                                       L.gs_sunkset_dev_loc(self._h), None, C.byref(h)))
        t = DeviceTable.__new__(DeviceTable)
        t._h, t.k, t.device, t.n = h, self.k, self.device, self.n
        return t

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_sunkset_dev_free(self._h)
            self._h = None

    __del__ = close


def device_reads_from_device_ascii(d_seq_ptr: int, off: np.ndarray, k: int, device: int = 0) -> DeviceReads:
    """gs_reads_from_device_ascii: reads whose bases (ASCII, back to back) are in device memory; packed on the GPU."""
    off = np.ascontiguousarray(off, np.uint64)
    h = C.c_void_p()
    check(lib().gs_reads_from_device_ascii(device, d_seq_ptr, off.ctypes.data, len(off) - 1, k, C.byref(h)))
    r = DeviceReads.__new__(DeviceReads)
    r._h, r.k, r.device = h, k, device
    r.n_reads, r.n_bases = len(off) - 1, int(off[-1])
    r.n_windows = lib().gs_reads_n_windows(h)
    return r
