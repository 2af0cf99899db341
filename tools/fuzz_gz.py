#!/usr/bin/env python3
"""Randomised cross-check of the native parallel gzip reader (csrc/gs_inflate.cpp) against zlib.  CPU only.
Every round: content of a random kind (sequence-like, FASTQ, text, runs, random bytes, mixtures), deflated at a random level /
strategy / window-less flush pattern, as one member or several, read back with a random thread count, piece size and request
size -- the bytes and the newline counts must be zlib's.  One round in four damages the file (a flipped bit, a cut) and expects
an error or, where zlib still accepts the file, zlib's bytes.
usage: fuzz_gz.py [seconds] [first seed]"""
import ctypes as C
import os
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import _lib  # noqa: E402

LUT = np.frombuffer(b"ACGT", np.uint8)


def content(rng):
    kind = int(rng.integers(0, 7))
    n = int(rng.choice([0, 1, 100, 70_000, 1_500_000, 9_000_000]))
    if kind == 0:
        return LUT[rng.integers(0, 4, n, dtype=np.uint8)].tobytes()
    if kind == 1:
        out, tot, i = [], 0, 0
        while tot < n:
            ln = int(rng.integers(1, 40000))
            out.append(b"@r%d\n" % i + LUT[rng.integers(0, 4, ln, dtype=np.uint8)].tobytes() + b"\n+\n" + b"#" * ln + b"\n")
            tot += 2 * ln
            i += 1
        return b"".join(out)
    if kind == 2:
        return b"".join(b"word%d and %d\n" % (int(a), int(b)) for a, b in rng.integers(0, 5000, (n // 16 + 1, 2)))[:n]
    if kind == 3:
        return b"".join(bytes([int(c)]) * int(r) for c, r in zip(rng.integers(0, 256, n // 300 + 1), rng.integers(1, 600, n // 300 + 1)))[:n]
    if kind == 4:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 5:  # a short alphabet with a skewed distribution: short literal codes
        return np.frombuffer(b"AAAAAAAACCCCGGT\n", np.uint8)[rng.integers(0, 16, n, dtype=np.uint8)].tobytes()
    a, b = content(rng), content(rng)
    return a[: len(a) // 2] + b + a[len(a) // 2:]


def deflate(rng, data):
    level = int(rng.choice([0, 1, 1, 3, 6, 6, 9]))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    memlevel = int(rng.choice([1, 4, 8, 9]))
    c = zlib.compressobj(level, zlib.DEFLATED, 31, memlevel, strategy)
    out = []
    if rng.random() < 0.3 and len(data) > 1000:
        step = int(rng.integers(500, max(len(data) // 3, 501)))
        for a in range(0, len(data), step):
            out.append(c.compress(data[a:a + step]))
            out.append(c.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK]))))
    else:
        out.append(c.compress(data))
    out.append(c.flush())
    return b"".join(out)


def zlib_bytes(blob):
    """What zlib's gzread semantics give: members one after the other, zero padding ignored; None when it refuses."""
    out, rest = [], blob
    try:
        while rest.strip(b"\0"):
            d = zlib.decompressobj(31)
            out.append(d.decompress(rest))
            if not d.eof:
                return None
            rest = d.unused_data
    except zlib.error:
        return None
    return b"".join(out)


def native(path, threads, want):
    L = _lib.lib()
    z = C.c_void_p()
    rc = L.gs_gz_open(path.encode(), threads, C.byref(z))
    if rc:
        return rc, b"", 0
    out, nl = [], 0
    try:
        while True:
            p, n = C.c_void_p(), C.c_uint64(0)
            rc = L.gs_gz_read(z, want, None, 0, C.byref(p), C.byref(n))
            if rc:
                return rc, b"", 0
            if n.value:
                out.append(C.string_at(p.value, n.value))
            nl += L.gs_gz_newlines(z)
            L.gs_gz_free_buffer(p)
            if not n.value or L.gs_gz_at_end(z):
                break
    finally:
        L.gs_gz_close(z)
    return 0, b"".join(out), nl


def one_round(seed, path):
    rng = np.random.default_rng(seed)
    members = [deflate(rng, content(rng)) for _ in range(int(rng.choice([1, 1, 1, 2, 5])))]
    blob = b"".join(members) + (b"\0" * int(rng.integers(0, 50)) if rng.random() < 0.2 else b"")
    damaged = rng.random() < 0.25 and len(blob) > 40
    if damaged:
        if rng.random() < 0.5:
            i = int(rng.integers(10, len(blob)))
            blob = blob[:i] + bytes([blob[i] ^ (1 << int(rng.integers(0, 8)))]) + blob[i + 1:]
        else:
            blob = blob[: int(rng.integers(10, len(blob)))]
    open(path, "wb").write(blob)
    want = zlib_bytes(blob)
    os.environ["GAVISUNK_GZ_PIECE_KB"] = str(int(rng.choice([64, 256, 2048])))
    rc, got, nl = native(path, int(rng.choice([1, 2, 5, 8])), int(rng.choice([1000, 1_000_000, 1 << 40])))
    if want is None:
        return None if rc != 0 else f"seed {seed}: zlib refuses the file, the reader returned {len(got)} bytes"
    if rc != 0:
        return None if damaged and not blob.startswith(b"\x1f\x8b") else f"seed {seed}: the reader refuses a file zlib reads ({len(want)} bytes, damaged={damaged})"
    if got != want or nl != want.count(b"\n"):
        return f"seed {seed}: {len(got)} bytes / {nl} newlines against zlib's {len(want)} / {want.count(10)}"
    return None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, rounds = time.time(), 0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "f.gz")
        while time.time() - t0 < budget:
            err = one_round(seed, path)
            if err:
                print("MISMATCH", err, flush=True)
                return 1
            rounds += 1
            seed += 1
            if rounds % 50 == 0:
                print(f"{rounds} rounds, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {rounds} rounds (seeds up to {seed - 1})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
