#!/usr/bin/env python3
"""Randomised cross-check of the on-device FASTA / FASTQ record parser against the host parser (GPU box; not part of the suite):
files of random size and shape (line width, CRLF, empty lines and records, lower case / N, block size of the reader), every
block parsed on the GPU, names / lengths / bases compared with gs_fastx_* on the same file.  usage: fuzz_ingest.py [seconds] [seed]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import _lib, engine  # noqa: E402

K = 20


def make(rng):
    n = int(rng.choice([1, 3, 40, 700]))
    fastq = rng.random() < 0.5
    crlf = rng.random() < 0.2
    width = int(rng.choice([0, 1, 7, 60, 80, 1000]))  # 0: one line per record
    alpha = np.frombuffer(b"ACGTacgtNnU", np.uint8)
    recs, seqs = [], []
    for i in range(n):
        ln = int(rng.choice([0, 1, 19, 20, 21, 300, 5000, 70000])) if not fastq else int(rng.choice([1, 19, 20, 300, 5000, 70000]))
        ln = int(rng.integers(max(ln // 2, 1 if fastq else 0), ln + 1))
        s = alpha[rng.integers(0, 4 if rng.random() < 0.8 else len(alpha), ln)].tobytes()
        seqs.append(s)
        name = b"r%d" % i + (b" extra\tfield" if rng.random() < 0.3 else b"")
        if fastq:
            q = bytes(rng.integers(33, 74, ln + int(rng.integers(0, 3)), dtype=np.uint8))
            recs.append(b"@" + name + b"\n" + s + b"\n+" + (name if rng.random() < 0.3 else b"") + b"\n" + q + b"\n")
        else:
            body = s if width == 0 else b"\n".join(s[j:j + width] for j in range(0, len(s), width))
            if rng.random() < 0.1:
                body += b"\n"  # an empty line inside the file
            recs.append(b">" + name + b"\n" + body + (b"\n" if len(body) or rng.random() < 0.5 else b""))
    text = b"".join(recs)
    if not text.endswith(b"\n") or rng.random() < 0.2:
        text = text.rstrip(b"\n")
    if rng.random() < 0.3:
        text += b"\n" * int(rng.integers(1, 4))
    if crlf:
        text = text.replace(b"\n", b"\r\n")
    return text


def one_file(seed, path):
    """One seeded file through both parsers -> "ok", "refused" (irregular: the host parser's job) -- or an AssertionError."""
    rng = np.random.default_rng(seed)
    text = make(rng)
    if not text:
        return "empty", 0
    open(path, "wb").write(text)
    blocks_seen = 0
    hn, hl, hbatch = [], [], []
    for b in engine.read_batches(path):
        hn += b.names
        hl += b.len.tolist()
        hbatch.append(b)
    dn, dl, dreads, ok = [], [], [], True
    for blk in engine.text_blocks(path, int(rng.choice([1 << 30, 1 << 16, 4096, 300]))):
        blocks_seen += 1
        try:
            r = engine.DeviceReads.from_text(blk, K)
        except _lib.GsError as e:
            assert e.status == _lib.GS_E_IRREGULAR, e
            ok = False
            break
        dn += r.names
        dl += r.len.tolist()
        dreads.append(r)
    try:
        if ok:
            assert dn == hn and dl == hl, (seed, len(dn), len(hn))
            # the bases: a table of this file's own k-mers; hit lists of both paths must be identical
            if sum(hl) >= K:
                from oracle import oracle
                keys = np.unique(np.concatenate([oracle.canonical_windows(bytes(s), K) for s in _seqs_of(text)]))[:200000]
                tab = engine.DeviceTable(K, keys, (np.arange(len(keys)) % 977).astype(np.uint32), np.arange(len(keys), dtype=np.uint32))
                carry, hh, base = _lib.GS_NONE, [], 0
                for b in hbatch:
                    dv = engine.DeviceReads(b, K)
                    _, h, carry, _ = engine.tag(tab, dv, carry)
                    h = h.copy(); h["read"] += base; base += b.n_reads
                    hh.append(h); dv.close()
                carry2, dh, base = _lib.GS_NONE, [], 0
                for r in dreads:
                    _, h, carry2, _ = engine.tag(tab, r, carry2)
                    h = h.copy(); h["read"] += base; base += r.n_reads
                    dh.append(h)
                assert carry == carry2 and np.array_equal(np.concatenate(hh), np.concatenate(dh)), seed
                tab.close()
    finally:
        for r in dreads:
            r.close()
        for b in hbatch:
            b.close()
    return ("ok" if ok else "refused"), blocks_seen


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, rounds, refused, blocks_seen = time.time(), 0, 0, 0
    d = tempfile.mkdtemp(prefix="gs_fuzz_ingest_")
    path = os.path.join(d, "f")
    while time.time() - t0 < budget:
        what, nb = one_file(seed, path)
        seed += 1
        if what == "empty":
            continue
        blocks_seen += nb
        refused += what == "refused"
        rounds += 1
    print(f"OK: {rounds} files (seeds up to {seed - 1}), {refused} refused as irregular, {blocks_seen} blocks")
    return 0


def _seqs_of(text):
    """sequences of a regular text, by the simplest possible reading (checker side)"""
    t = text.replace(b"\r\n", b"\n")
    out = []
    if t[:1] == b">":
        for rec in t.split(b"\n>"):
            lines = rec.split(b"\n")
            s = b"".join(lines[1:])
            out.append(s if len(s) >= K else s + b"A" * (K - len(s)))
    else:
        lines = t.split(b"\n")
        for i in range(1, len(lines), 4):
            s = lines[i]
            out.append(s if len(s) >= K else s + b"A" * (K - len(s)))
    return out


if __name__ == "__main__":
    sys.exit(main())
