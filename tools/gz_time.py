#!/usr/bin/env python3
"""Phase times of the native parallel gzip reader (csrc/gs_inflate.cpp) on a genome-scale-like `.fq.gz` (multi-member, quals '#'),
by thread count.  CPU only (run it on the GPU box for that box's host cores).  usage: gz_time.py [GB of text, default 6]"""
import ctypes as C
import gzip
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import _lib  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
path = "/tmp/gz_time.fq.gz"
n_members = int(gb * 1e9 / 40e6)
lut = np.frombuffer(b"ACGT", np.uint8)


def member(i):
    rng = np.random.default_rng(i)
    out = []
    for j in range(500):
        ln = int(rng.integers(20000, 60000))
        out.append(b"@m%d_%d\n" % (i, j) + lut[rng.integers(0, 4, ln, dtype=np.uint8)].tobytes() + b"\n+\n" + b"#" * ln + b"\n")
    return gzip.compress(b"".join(out), 1)


t0 = time.time()
with open(path, "wb") as f, ThreadPoolExecutor(16) as pool:
    for blob in pool.map(member, range(n_members)):
        f.write(blob)
print(f"{path}: {os.path.getsize(path) / 2**30:.2f} GiB in {n_members} members ({time.time() - t0:.0f} s); hardware threads {os.cpu_count()}, "
      f"affinity {len(os.sched_getaffinity(0))}", flush=True)
L = _lib.lib()
for threads, hold in [(int(x), 0) for x in os.environ.get("GZ_TIME_THREADS", "").split(",") if x] or ((8, 0), (16, 0), (32, 0), (32, 5), (32, 5)):
    # hold: text blocks the consumer still holds when the next one is made (the command line holds up to 5: being inflated, queued,
    # being copied, queued, being parsed) -- those cannot come back from the reader's buffer pool
    z = C.c_void_p()
    _lib.check(L.gs_gz_open(path.encode(), threads, C.byref(z)))
    t0 = time.time()
    total, held = 0, []
    while True:
        p, n = C.c_void_p(), C.c_uint64(0)
        _lib.check(L.gs_gz_read(z, 1 << 30, None, 0, C.byref(p), C.byref(n)))
        total += n.value
        held.append(p)
        while len(held) > hold:
            L.gs_gz_free_buffer(held.pop(0))
        if not n.value or L.gs_gz_at_end(z):
            break
    dt = time.time() - t0
    for p in held:
        L.gs_gz_free_buffer(p)
    L.gs_gz_close(z)
    print(f"threads {threads:2d}, {hold} blocks held: {total / 1e9:.2f} GB of text in {dt:.2f} s = {total / dt / 1e9:.2f} GB/s", flush=True)
if not os.environ.get("GZ_TIME_KEEP"):
    os.unlink(path)
