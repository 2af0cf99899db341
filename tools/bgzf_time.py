#!/usr/bin/env python3
"""The GPU BGZF reader alone (csrc/gs_bgzf.hip): GB/s of text for a FASTQ file shaped like `seqtk seq -F '#' | rustybam fastq-split`
output (bgzip members of 64 KiB).  usage: bgzf_time.py [GB of text, default 4] [level, default 6]"""
import os
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import engine  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(1)
lut = np.frombuffer(b"ACGT", np.uint8)


def member(chunk):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    raw = c.compress(chunk) + c.flush()
    return (b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1) + raw +
            struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def piece(seed):
    r = np.random.default_rng(seed)
    recs = []
    for i in range(400):
        n = int(r.lognormal(np.log(30000), 0.6))
        recs.append(b"@g%d_%d\n" % (seed, i) + lut[r.integers(0, 4, n, dtype=np.uint8)].tobytes() + b"\n+\n" + b"#" * n + b"\n")
    return b"".join(recs)


with tempfile.TemporaryDirectory(prefix="gs_bgzf_", dir="/tmp") as d:
    path = f"{d}/x.fq.gz"
    total = 0
    with open(path, "wb") as f, ThreadPoolExecutor(16) as pool:
        seed = 0
        carry = b""
        while total < gb * 1e9:
            texts = list(pool.map(piece, range(seed, seed + 16)))
            seed += 16
            text = carry + b"".join(texts)
            cut = len(text) // 65280 * 65280
            carry, text = text[cut:], text[:cut]
            for blob in pool.map(member, [text[a:a + 65280] for a in range(0, len(text), 65280)]):
                f.write(blob)
            total += len(text)
        f.write(member(carry))
        f.write(member(b""))
        total += len(carry)
    print(f"{total / 1e9:.2f} GB of text, {os.path.getsize(path) / 2**30:.2f} GiB compressed (level {level})", flush=True)
    for rep in range(3):
        t0 = time.perf_counter()
        z = engine.BgzfReader(path)
        n = 0
        for st in z.blocks(int(os.environ.get("BGZF_BLOCK_MB", "3000")) << 20):
            n += st.host_copy().__len__() if rep == 0 and os.environ.get("BGZF_CHECK") else 0
            st.close()
        z.close()
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {dt:.3f} s = {total / dt / 1e9:.1f} GB/s of text ({os.path.getsize(path) / dt / 1e9:.1f} GB/s compressed)", flush=True)
