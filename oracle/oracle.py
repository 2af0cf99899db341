"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of GAVISUNK's hot path, used only as the parity checker by tests/,
`__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg.  `gavisunk_amd/` never imports it.

  tagger          gs_oracle.c (C, via ctypes)   <- workflow/src/kmerpos_annot3.nim
  rlen            read_lengths()                <- workflow/src/rlen.nim:13-14
  diag filter     diag_filter_v3()              <- workflow/src/diag_filter_v3.nim:30-229
  diag step 2     diag_filter_step2()           <- workflow/src/diag_filter_step2.nim:16-64
  pair/component  pair_component_read(), process_by_contig()
                                                <- workflow/scripts/process-by-contig_lowmem_AR.py:50-260

Parity status
  tagger, rlen, diag filter, step 2: PINNED against outputs of the reference's own prebuilt
    binaries (tests/golden/, made by tests/golden/make_golden.py).
  pair/component stage: PROVISIONALLY PINNED -- the reference script needs graph_tool / seaborn /
    pandas 1.3.4, none of which exist in the build image, and the reference ships no expected
    output for it.  tests/golden/pair_* hold outputs of the reference's own script run here under
    pandas 2.x with a documented graph_tool stand-in (tests/golden/make_golden_pair.py): everything up
    to the graph calls is the reference's code executing for real; vertex order / largest-component
    tie rules are the stand-in's.  Exact wherever no tie is involved, unpinned for ties.
  bad-SUNK caller: PARITY UNPINNED (the script does not run under pandas 2.x); restated from the text.
"""
from __future__ import annotations

import ctypes
import gzip
import itertools
import math
import os
import subprocess
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(HERE, "_build", "libgsoracle.so")
NONE = 0xFFFFFFFF


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "gs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.gso_encode.restype = ctypes.c_uint64
        L.gso_encode.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.gso_table_new.restype = ctypes.c_void_p
        L.gso_table_new.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.gso_table_free.argtypes = [ctypes.c_void_p]
        L.gso_tag.restype = ctypes.c_int
        L.gso_tag.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        L.gso_canonical_windows.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
        _lib = L
    return _lib


HIT_DTYPE = np.dtype([("read", "<u4"), ("pos", "<u4"), ("row", "<u4")])


# ------------------------------------------------------------------------------ file parsing
def _open(path: str):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def parse_reads(path: str) -> Tuple[List[str], List[bytes]]:
    """kseq semantics (the reference reads through `readfq`, kmerpos_annot3.nim:85): FASTA or
    FASTQ, plain or gzip, multi-line; name = header up to the first whitespace."""
    names, seqs = [], []
    with _open(path) as f:
        data = f.read()
    lines = data.split(b"\n")
    i, n = 0, len(lines)
    while i < n:
        ln = lines[i]
        if not ln or ln[:1] not in (b">", b"@"):
            i += 1
            continue
        hdr = ln[1:].split(None, 1)
        names.append(hdr[0].decode() if hdr else "")
        i += 1
        chunks = []
        while i < n and lines[i][:1] not in (b">", b"@", b"+"):
            chunks.append(lines[i].rstrip(b"\r"))
            i += 1
        seq = b"".join(chunks)
        seqs.append(seq)
        if i < n and lines[i][:1] == b"+":  # FASTQ: skip qualities (as many bytes as bases)
            i += 1
            q = 0
            while i < n and q < len(seq):
                q += len(lines[i].rstrip(b"\r"))
                i += 1
    return names, seqs


def read_lengths(path: str) -> str:
    """rlen.nim:13-14: `name \\t len(sequence)` per record."""
    names, seqs = parse_reads(path)
    return "".join(f"{n}\t{len(s)}\n" for n, s in zip(names, seqs))


def load_db(path: str) -> Tuple[int, np.ndarray]:
    """kmerpos_annot3.nim:20-38: one k-mer per line, k = len(first line)."""
    with _open(path) as f:
        lines = f.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    k = len(lines[0]) if lines else 0
    L = lib()
    codes = np.fromiter((L.gso_encode(s, len(s)) for s in lines), dtype=np.uint64, count=len(lines))
    return k, codes


def load_loc(path: str):
    """kmerpos_annot3.nim:53-69: TSV chrom, start, kmer, group."""
    chroms: Dict[str, int] = {}
    names: List[str] = []
    cid, start, code, group = [], [], [], []
    L = lib()
    with _open(path) as f:
        for ln in f:
            ln = ln.rstrip(b"\n")
            if not ln:
                continue
            c, s, kmer, g = ln.split(b"\t")[:4]
            c = c.decode()
            if c not in chroms:
                chroms[c] = len(names)
                names.append(c)
            cid.append(chroms[c])
            start.append(int(s))
            code.append(L.gso_encode(kmer, len(kmer)))
            group.append(int(g))
    return (names, np.array(cid, dtype=np.uint32), np.array(start, dtype=np.int64),
            np.array(code, dtype=np.uint64), np.array(group, dtype=np.int64))


# ------------------------------------------------------------------------------ tagger
class Table:
    def __init__(self, k: int, db_codes: np.ndarray, loc_codes: np.ndarray, loc_chrom: np.ndarray,
                 loc_group: np.ndarray):
        self.k = k
        self._keep = [np.ascontiguousarray(db_codes, np.uint64), np.ascontiguousarray(loc_codes, np.uint64),
                      np.ascontiguousarray(loc_chrom, np.uint32), np.ascontiguousarray(loc_group, np.int64)]
        a = self._keep
        self._h = lib().gso_table_new(k, len(a[0]), a[0].ctypes.data, len(a[1]), a[1].ctypes.data,
                                      a[2].ctypes.data, a[3].ctypes.data)
        if not self._h:
            raise MemoryError

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gso_table_free(self._h)
            self._h = None

    def tag(self, seq: np.ndarray, read_off: np.ndarray, prev_row: int = NONE, cap: int | None = None):
        """-> (rc, hits[HIT_DTYPE], prev_row_out, counters dict).  rc 0 ok, -2 KeyError."""
        seq = np.ascontiguousarray(seq, np.uint8)
        read_off = np.ascontiguousarray(read_off, np.uint64)
        R = len(read_off) - 1
        if cap is None:
            cap = max(16, len(seq))
        out = np.zeros(cap, dtype=HIT_DTYPE)
        prev = ctypes.c_uint32(prev_row)
        n = ctypes.c_uint64(0)
        ctr = (ctypes.c_uint64 * 3)()
        rc = lib().gso_tag(self._h, seq.ctypes.data, read_off.ctypes.data, R, ctypes.byref(prev),
                           out.ctypes.data, cap, ctypes.byref(n), ctr)
        return rc, out[: n.value].copy(), prev.value, {"windows": ctr[0], "raw_hits": ctr[1], "emitted": ctr[2]}


def concat_reads(seqs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(seqs), dtype=np.uint8) if seqs else np.zeros(0, np.uint8)
    return buf, off


def tag_files(reads: str, db: str, loc: str) -> Tuple[int, str]:
    """`kmerpos_annot3 <reads> <db> <loc> <out>` -> (exit code, contents of <out>)."""
    k, dbc = load_db(db)
    cnames, cid, start, code, group = load_loc(loc)
    t = Table(k, dbc, code, cid, group)
    names, seqs = parse_reads(reads)
    buf, off = concat_reads(seqs)
    rc, hits, _, _ = t.tag(buf, off)
    txt = format_sunkpos(hits, names, cnames, cid, start, group)
    return (1 if rc == -2 else 0), txt


def tag_files_chunked(reads: str, db: str, loc: str, nchunks: int) -> Tuple[int, str]:
    """The workflow's scatter (tagONT.smk:2-37,112-131): records dealt round-robin to `nchunks` files, the tagger run on each
    with a fresh `prevLoc`, outputs concatenated in scatter order."""
    k, dbc = load_db(db)
    cnames, cid, start, code, group = load_loc(loc)
    t = Table(k, dbc, code, cid, group)
    names, seqs = parse_reads(reads)
    out, code_rc = [], 0
    for c in range(nchunks):
        buf, off = concat_reads(seqs[c::nchunks])
        rc, hits, _, _ = t.tag(buf, off)
        out.append(format_sunkpos(hits, names[c::nchunks], cnames, cid, start, group))
        code_rc = code_rc or (1 if rc == -2 else 0)
    return code_rc, "".join(out)


def format_sunkpos(hits, names, cnames, cid, start, group) -> str:
    """kmerpos_annot3.nim:93: name, i, chrom, start, group."""
    return "".join(f"{names[h['read']]}\t{h['pos']}\t{cnames[cid[h['row']]]}\t{start[h['row']]}\t{group[h['row']]}\n"
                   for h in hits)


def canonical_windows(seq: bytes, k: int) -> np.ndarray:
    a = np.frombuffer(seq, dtype=np.uint8)
    n = max(0, len(a) - k + 1)
    out = np.zeros(n, dtype=np.uint64)
    if n:
        lib().gso_canonical_windows(a.ctypes.data, len(a), k, out.ctypes.data)
    return out


# ------------------------------------------------------------------------------ Nim Table[string, ...] slot order
def murmur3_32(data: bytes) -> int:
    """Nim >= 1.4 `hash(string)` = MurmurHash3_x86_32, seed 0 (symbol `murmurHash_pureZhashes`
    in the reference's diag_filter_v3 ELF)."""
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = 0
    n = len(data) // 4
    for i in range(n):
        k = int.from_bytes(data[4 * i:4 * i + 4], "little")
        k = (k * c1) & 0xFFFFFFFF
        k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
        h = ((h << 13) | (h >> 19)) & 0xFFFFFFFF
        h = (h * 5 + 0xE6546B64) & 0xFFFFFFFF
    tail = data[4 * n:]
    k = int.from_bytes(tail, "little") if tail else 0
    k = (k * c1) & 0xFFFFFFFF
    k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
    k = (k * c2) & 0xFFFFFFFF
    h ^= k
    h ^= len(data)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


class NimStringTable:
    """Slot layout of Nim's `Table[string, V]` (open addressing, linear probing, 64 initial
    slots for `initTable()`, grows x2 when `dataLen*2 < counter*3 or dataLen-counter < 4`
    before an insert; `clear` keeps the capacity).  Only what `pairs()` order needs."""

    def __init__(self):
        self.slots: List[str | None] = [None] * 64
        self.hc: List[int] = [0] * 64
        self.counter = 0

    @staticmethod
    def _hash(key: str) -> int:
        h = murmur3_32(key.encode())
        return h if h != 0 else 314159265

    def clear(self):
        n = len(self.slots)
        self.slots, self.hc, self.counter = [None] * n, [0] * n, 0

    def _find(self, slots, key, hc):
        mask = len(slots) - 1
        h = hc & mask
        while slots[h] is not None:
            if slots[h] == key:
                return h, True
            h = (h + 1) & mask
        return h, False

    def add(self, key: str) -> None:
        hc = self._hash(key)
        h, found = self._find(self.slots, key, hc)
        if found:
            return
        n = len(self.slots)
        if n * 2 < self.counter * 3 or n - self.counter < 4:
            new, newhc = [None] * (2 * n), [0] * (2 * n)
            for i in range(n):
                if self.slots[i] is not None:
                    j, _ = self._find(new, self.slots[i], self.hc[i])
                    new[j], newhc[j] = self.slots[i], self.hc[i]
            self.slots, self.hc = new, newhc
            h, _ = self._find(self.slots, key, hc)
        self.slots[h], self.hc[h] = key, hc
        self.counter += 1

    def keys(self) -> List[str]:
        return [s for s in self.slots if s is not None]


# ------------------------------------------------------------------------------ diag filter
def _percentile50_trunc(vals: List[int]) -> int:
    """arraymancer 0.7.10 `percentile(t, 50)` then Nim `int()` (diag_filter_v3.nim:86,113):
    sort, linear interpolation at (n-1)*0.5 in float64, truncate toward zero."""
    if not vals:
        return 0
    a = sorted(vals)
    f = (len(a) - 1) * 50 / 100
    i = math.floor(f)
    frac = f - i
    if f == float(i):
        res = float(a[i])
    else:
        res = float(a[i]) + frac * (float(a[i + 1]) - float(a[i]))
    return int(res)  # trunc toward zero


def parse_sunkpos(text: str) -> List[Tuple[str, int, str, int, int]]:
    rows = []
    for ln in text.split("\n"):
        if not ln:
            continue
        f = ln.split("\t")
        rows.append((f[0], int(f[1]), f[2], int(f[3]), int(f[4])))
    return rows


def diag_filter_v3(rows: Sequence[Tuple[str, int, str, int, int]], fai_contigs: Iterable[str],
                   bandwidth: int = 2500) -> str:
    """diag_filter_v3.nim:63-229.  rows = (rname, pos, contig, start, group) in file order."""
    contigs = set(fai_contigs)  # :30-37
    table = NimStringTable()    # posStarts; one table reused, clear() after each read (:143)
    data: Dict[str, Tuple[List[int], List[int], List[int]]] = {}
    out = []

    def flush(rname):
        maxgood, bestcontig, bestdir, maxhitlen = 0, "", "", 0
        for c in table.keys():  # Table slot order (:76)
            p, s, g = data[c]
            hitlen = len(g)
            if len(p) == 1:     # :82
                continue
            for direction in ("-", "+"):  # "-" (p+s) first (:84), then "+" (s-p) (:111)
                n = [pi + si for pi, si in zip(p, s)] if direction == "-" else [si - pi for pi, si in zip(p, s)]
                median = _percentile50_trunc(n)
                good = {gi for v, gi in zip(n, g) if abs(v - median) < bandwidth}  # strict (:91)
                if len(good) > maxgood:                             # :98-102
                    maxgood, maxhitlen, bestcontig, bestdir = len(good), hitlen, c, direction
                if len(good) == maxgood and hitlen < maxhitlen:     # :103-107
                    maxgood, maxhitlen, bestcontig, bestdir = len(good), hitlen, c, direction
        if maxgood > 1:  # :141
            out.append(f"{rname}\t{bestcontig}\t{maxgood}\t{bestdir}\t{maxgood}\n")

    prev = ""
    for rname, pos, contig, start, group in rows:
        if rname != prev:  # :64 (fires for the very first row too, with an empty table)
            flush(prev)
            table.clear()
            data = {}
        if contig not in data:  # :145-146 -- entry is created even for contigs outside the fai
            table.add(contig)
            data[contig] = ([], [], [])
        if contig in contigs:   # :147-150
            data[contig][0].append(pos)
            data[contig][1].append(start)
            data[contig][2].append(group)
        prev = rname
    flush(prev)  # :156-229
    return "".join(out)


def diag_filter_step2(sunkpos_text: str, diag_text: str) -> str:
    """diag_filter_step2.nim:16-64: keep rows whose contig is the read's best contig."""
    best = {}
    for ln in diag_text.split("\n"):
        if ln:
            f = ln.split("\t")
            best[f[0]] = f[1]
    out = []
    for ln in sunkpos_text.split("\n"):
        if not ln:
            continue
        f = ln.split("\t")
        if f[2] == best.get(f[0], ""):
            out.append(ln + "\n")
    return "".join(out)


# ------------------------------------------------------------------------------ pair / component stage
def pair_component_read_np(ids: Sequence[int], pos: Sequence[int], start: Sequence[int]) -> List[int]:
    """pair_component_read below with NumPy / SciPy arrays instead of Python loops -- for reads with thousands of hits (the
    loop form needs minutes beyond n ~ 10^3).  The same reference lines; tests/test_pair_golden.py holds the two against each
    other on every fixture before the GPU tests rely on this one."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    ids = np.asarray(ids, dtype=np.int64)
    posa = np.asarray(pos, dtype=np.int64)
    sta = np.asarray(start, dtype=np.int64)
    n = len(ids)
    if n < 2:
        return []
    ii, jj = np.triu_indices(n, 1)                                   # pdist / combinations order (:140-144)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.abs(posa[ii].astype(np.float64) - posa[jj].astype(np.float64)) / np.abs(sta[ii].astype(np.float64) - sta[jj].astype(np.float64))
    mask = ~((ratio >= 1.1) | (ratio <= 0.9))                        # :147 (NaN passes)
    if not mask.any():                                               # :148
        return []
    sign = posa[ii] > posa[jj]                                        # :142
    ones, zeros = int((mask & sign).sum()), int((mask & ~sign).sum())
    keep = mask & (sign == (ones > zeros))                           # :150-153, tie -> 0
    ei, ej = ii[keep], jj[keep]
    E = len(ei)
    # mergedf = concat(ID/pos1 rows, then ID2/pos2 rows) (:160-163); per ID the most frequent pos, first seen on ties
    m_id = np.concatenate([ids[ei], ids[ej]])
    m_pos = np.concatenate([posa[ei], posa[ej]])
    # (ID, pos) pairs as one integer where they fit (np.unique over rows is an order of magnitude slower)
    if len(m_id) and 0 <= m_id.min() and m_id.max() < (1 << 31) and 0 <= m_pos.min() and m_pos.max() < (1 << 32):
        packed, first, inv, cnt = np.unique((m_id << 32) | m_pos, return_index=True, return_inverse=True, return_counts=True)
        uniq = np.stack([packed >> 32, packed & 0xFFFFFFFF], axis=1)
    else:
        uniq, first, inv, cnt = np.unique(np.stack([m_id, m_pos], axis=1), axis=0, return_index=True, return_inverse=True, return_counts=True)
    uid, uinv = np.unique(uniq[:, 0], return_inverse=True)
    multi_id = uid[np.bincount(uinv.reshape(-1)) > 1]
    bad = np.zeros(E, dtype=bool)
    if len(multi_id):
        # best (count desc, first occurrence asc) pair per ID
        order = np.lexsort((first, -cnt, uniq[:, 0]))
        best_of = {}
        for q in order:
            best_of.setdefault(int(uniq[q, 0]), int(uniq[q, 1]))
        goodpos_of_edge = np.array([best_of[int(x)] for x in ids[ei]], dtype=np.int64)
        is_multi = np.isin(ids[ei], multi_id)
        bad = is_multi & ~((posa[ei] == goodpos_of_edge) | (posa[ej] == goodpos_of_edge))   # :175-176 (first ID only)
    ei, ej = ei[~bad], ej[~bad]
    if not len(ei):
        return []
    # vertices = ID values numbered by first appearance, source then target, edge by edge (:189-198)
    seq = np.stack([ids[ei], ids[ej]], axis=1).reshape(-1)
    vals, first_seen = np.unique(seq, return_index=True)
    vorder = np.argsort(first_seen, kind="stable")
    verts = vals[vorder]
    vid_of_val = np.empty(len(vals), dtype=np.int64)
    vid_of_val[vorder] = np.arange(len(vals))
    a = vid_of_val[np.searchsorted(vals, ids[ei])]
    b = vid_of_val[np.searchsorted(vals, ids[ej])]
    nv = len(verts)
    ncomp, lab = connected_components(coo_matrix((np.ones(len(a), np.int8), (a, b)), shape=(nv, nv)), directed=False)
    # label_components labels in order of first vertex; largest = first arg-max over labels
    _, first_v = np.unique(lab, return_index=True)
    relabel = np.empty(ncomp, dtype=np.int64)
    relabel[np.argsort(first_v, kind="stable")] = np.arange(ncomp)
    lab = relabel[lab]
    best = int(np.argmax(np.bincount(lab, minlength=ncomp)))
    return [int(v) for v in verts[lab == best]]


def pair_component_read(ids: Sequence[int], pos: Sequence[int], start: Sequence[int]) -> List[int]:
    """One read of process-by-contig_lowmem_AR.py:135-198.  Inputs are the read's rows after the
    pre-filters, sorted by `start` (stable), exact duplicates dropped.  Returns the IDs of the
    largest connected component in vertex insertion order, or [] when the read is skipped."""
    n = len(ids)
    pairs = list(itertools.combinations(range(n), 2))  # pdist / combinations order (:140-144)
    if not pairs:
        return []
    mask, sign = [], []
    for i, j in pairs:
        sd = abs(float(start[i]) - float(start[j]))
        pd_ = abs(float(pos[i]) - float(pos[j]))
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.float64(pd_) / np.float64(sd)               # :146
        mask.append(not (ratio >= 1.1 or ratio <= 0.9))            # :147 (NaN passes)
        sign.append(1 if pos[i] > pos[j] else 0)                   # :142
    if sum(mask) < 1:                                              # :148
        return []
    ones = sum(1 for m, s in zip(mask, sign) if m and s == 1)
    zeros = sum(1 for m, s in zip(mask, sign) if m and s == 0)
    true_orient = 1 if ones > zeros else 0                         # :150-152 argmax over sorted [0, 1], tie -> 0
    edges = [(ids[i], ids[j], pos[i], pos[j]) for (i, j), m, s in zip(pairs, mask, sign)
             if m and s == true_orient]                            # :153-158
    # multipos clean-up (:160-181)
    seen: Dict[int, Dict[int, int]] = {}
    order: List[Tuple[int, int]] = []
    for a, b, pa, pb in edges:       # mergedf = concat(ID/pos1 rows, then ID2/pos2 rows)
        order.append((a, pa))
    for a, b, pa, pb in edges:
        order.append((b, pb))
    for i_, p_ in order:
        seen.setdefault(i_, {})
        seen[i_][p_] = seen[i_].get(p_, 0) + 1
    bad = set()
    for i_ in sorted(k_ for k_, v in seen.items() if len(v) > 1):  # groupby index is sorted
        cnt = seen[i_]
        best = max(cnt.values())
        # value_counts().index[0]: most frequent; ties -> first encountered (pandas >= 1.? keeps
        # first-seen order among equal counts; pandas-version dependent)
        goodpos = next(p_ for (ii, p_) in order if ii == i_ and cnt[p_] == best)
        for e, (a, b, pa, pb) in enumerate(edges):
            if a == i_ and not (pa == goodpos or pb == goodpos):   # :175-176 (first ID only)
                bad.add(e)
    edges2 = [e for i_, e in enumerate(edges) if i_ not in bad]
    # graph, largest component, vertex insertion order (:189-198)
    vid: Dict[int, int] = {}
    verts: List[int] = []
    parent: List[int] = []

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b, _, _ in edges2:
        for v in (a, b):
            if v not in vid:
                vid[v] = len(verts)
                verts.append(v)
                parent.append(len(parent))
        ra, rb = find(vid[a]), find(vid[b])
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    if not verts:
        return []
    roots = [find(i) for i in range(len(verts))]
    size: Dict[int, int] = {}
    for r in roots:
        size[r] = size.get(r, 0) + 1
    # label_components labels in order of first vertex; largest = first arg-max over labels
    label_order = []
    for r in roots:
        if r not in label_order:
            label_order.append(r)
    best_root = max(label_order, key=lambda r: (size[r], -label_order.index(r)))
    return [verts[i] for i in range(len(verts)) if roots[i] == best_root]


def per_read_inputs(sunkpos_text: str, rlen_text: str, bad_text: str):
    """The file-level filtering of process-by-contig_lowmem_AR.py:50-108 -> (contig, {rname: rows}) with a read's rows sorted
    by `start` (stable) and de-duplicated; None when nothing survives (:92-94)."""
    rows = parse_sunkpos(sunkpos_text)                                   # :60
    contig = rows[0][2]                                                  # :62
    bad = set(bad_text.splitlines())                                     # :67-68
    rows = [r for r in rows if f"{r[2]}:{r[4]}" not in bad]              # :69-72
    ids_of: Dict[str, set] = {}
    for r in rows:
        ids_of.setdefault(r[0], set()).add(r[4])
    multi = {n for n, s in ids_of.items() if len(s) > 1}                 # :91
    if not multi:
        return contig, None                                              # :92-94
    rows = [r for r in rows if r[0] in multi]
    rows.sort(key=lambda r: (r[0], r[3]))                                # :100 (stable)
    seen, ded = set(), []
    for r in rows:                                                       # :104
        if r not in seen:
            seen.add(r)
            ded.append(r)
    rlen = {}
    for ln in rlen_text.splitlines():
        f = ln.split("\t")
        if len(f) >= 2:
            rlen[f[0]] = int(f[1])
    sub = [r for r in ded if rlen.get(r[0], -1) >= 10000]                # :106-108
    groups: Dict[str, list] = {}
    for r in sub:
        groups.setdefault(r[0], []).append(r)
    return contig, groups


def process_by_contig(sunkpos_text: str, rlen_text: str, bad_text: str) -> Tuple[str, str]:
    """process-by-contig_lowmem_AR.py:50-260 on the contents of its input files -> (tsv, bed).
    Plain-Python restatement (dicts and lists instead of pandas); PARITY UNPINNED, see module header."""
    contig, groups = per_read_inputs(sunkpos_text, rlen_text, bad_text)
    if groups is None:
        return f"{contig}\n", ""                                         # :92-94
    outputs = []
    for rname in sorted(groups):                                         # :135 groupby -> sorted keys
        g = groups[rname]
        ids = pair_component_read([r[4] for r in g], [r[1] for r in g], [r[3] for r in g])
        if ids:
            outputs.append((rname, ids))
    if not outputs:
        return f"{contig}\n", ""                                         # :202-204
    tsv = "".join(f"{i}\t{rname}\n" for rname, ids in outputs for i in ids)  # :206-207
    # contig-wide graph (:219-260)
    vid: Dict[int, int] = {}
    verts: List[int] = []
    parent: List[int] = []

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for _, ids in outputs:
        for a, b in itertools.combinations(ids, 2):
            for v in (a, b):
                if v not in vid:
                    vid[v] = len(verts)
                    verts.append(v)
                    parent.append(len(parent))
            ra, rb = find(vid[a]), find(vid[b])
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    comps: Dict[int, list] = {}
    for i, v in enumerate(verts):
        comps.setdefault(find(i), []).append(v)
    regs = sorted((min(c), max(c)) for c in comps.values() if len(c) > 2)
    bed = "".join(f"{contig}\t{s}\t{e}\n" for s, e in regs)
    return tsv, bed


def badsunks(fai1_text: str, fai2_text: str, sunkpos1_text: str, sunkpos2_text: str) -> set:
    """badsunks_AR.py:20-103, restated from the source text.  Returns the set the script writes (in arbitrary order) to <outpath>.
    Pinned by tests/golden/badsunks/*: the script's own outputs, run in place under pandas 2.x through a two-function adapter that
    restores the pandas-1.3 naming it was written for (tests/golden/_shim/pandas13_names.py; tests/test_oracle_golden.py)."""
    from collections import Counter

    def one(fai_text, sp_text):
        contigs = {ln.split("\t")[0] for ln in fai_text.splitlines() if ln.strip()}
        cnt = Counter(f"{r[2]}:{r[4]}" for r in parse_sunkpos(sp_text))       # :22-24
        own = {k: v for k, v in cnt.items() if k.split(":", 1)[0] in contigs}     # :29-34 `str.split(":", 1)`: the text
        other = {k: v for k, v in cnt.items() if k not in own}                    #  before the FIRST colon is looked up
        hist = Counter(own.values())
        top = max(hist.values())
        meancov = min(c for c, n in hist.items() if n == top)                     # :43 mode().values[0]
        limit = meancov + (meancov ** 0.5) * 4                                    # :46
        return ({k for k, v in own.items() if v > limit or v < 2} |               # :48
                {k for k, v in other.items() if v > limit})                       # :51
    return one(fai1_text, sunkpos1_text) | one(fai2_text, sunkpos2_text)          # :97
