#!/usr/bin/env python3
"""Repeats the command line on one BGZF-shaped .fq.gz (tools/cli_genome_time.py's file, kept between runs) with a watchdog that dumps
every thread's stack if a run takes longer than a minute.  usage: cli_bgzf_repeat.py [runs] [reads]"""
import faulthandler
import os
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import engine, kmerpos_annot3, synth_gpu  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
with tempfile.TemporaryDirectory(prefix="gs_clirep_", dir="/tmp") as d:
    gen = synth_gpu.Genome(200_000_000, n_contigs=24, seed=31, device=0)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, 20, 0)
    sunks.write_files(gen.names, f"{d}/db", f"{d}/loc")
    rng = np.random.default_rng(31)
    big = synth_gpu.ReadSet(gen, synth_gpu.lognormal_lengths(rng, n_reads, median=32000.0), seed=31001)
    gen.close()
    sunks.close()
    nb, bases = big.n_reads, big.n_bases

    def member(lo):
        hi = min(lo + 500, nb)
        text = b"".join(b"@g%d\n" % (lo + j) + sq + b"\n+\n" + b"#" * len(sq) + b"\n" for j, sq in enumerate(big.host_reads(lo, hi)))
        out = []
        for a in range(0, len(text), 65280):
            chunk = text[a:a + 65280]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            raw = c.compress(chunk) + c.flush()
            out.append(b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1) + raw +
                       struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
        return b"".join(out)

    with open(f"{d}/big.fq.gz", "wb") as f, ThreadPoolExecutor(16) as pool:
        for blob in pool.map(member, range(0, nb, 500)):
            f.write(blob)
    big.close()
    assert kmerpos_annot3.main(["--prepare", f"{d}/db", f"{d}/loc"]) == 0
    print(f"{nb} reads, {bases} bases, big.fq.gz {os.path.getsize(f'{d}/big.fq.gz') / 2**30:.2f} GiB", flush=True)
    for r in range(runs):
        faulthandler.dump_traceback_later(60, exit=True)
        t0 = time.perf_counter()
        rc = kmerpos_annot3.main([f"{d}/big.fq.gz", f"{d}/db", f"{d}/loc", f"{d}/out"])
        dt = time.perf_counter() - t0
        faulthandler.cancel_dump_traceback_later()
        tm = {k: round(v, 2) if isinstance(v, float) else v for k, v in kmerpos_annot3.last_timing.items()}
        print(f"run {r}: rc {rc}, {dt:.2f} s = {bases / dt / 1e9:.2f} Gbase/s, phases {tm}", flush=True)
