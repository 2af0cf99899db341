#!/usr/bin/env python3
"""The command line at genome scale (bench.py's `cli_genome_scale` leg: 250 000 reads = 1e10 bases as a 3.25 GiB multi-member
.fq.gz against the 1e8-key surrogate's sidecar), seconds by phase for several inflate thread counts (GAVISUNK_GZ_THREADS).  GPU box."""
import gzip
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import engine, kmerpos_annot3, synth_gpu  # noqa: E402

with tempfile.TemporaryDirectory(prefix="gs_cligen_", dir="/tmp") as d:
    gen = synth_gpu.Genome(200_000_000, n_contigs=24, seed=31, device=0)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, 20, 0)
    sunks.write_files(gen.names, f"{d}/db", f"{d}/loc")
    rng = np.random.default_rng(31)
    big = synth_gpu.ReadSet(gen, synth_gpu.lognormal_lengths(rng, int(sys.argv[1]) if len(sys.argv) > 1 else 250_000, median=32000.0), seed=31001)
    gen.close()
    sunks.close()
    nb, bases = big.n_reads, big.n_bases

    def member(lo):
        hi = min(lo + 500, nb)
        return gzip.compress(b"".join(b"@g%d\n" % (lo + j) + sq + b"\n+\n" + b"#" * len(sq) + b"\n" for j, sq in enumerate(big.host_reads(lo, hi))), 1)

    with open(f"{d}/big.fq.gz", "wb") as f, ThreadPoolExecutor(16) as pool:
        for blob in pool.map(member, range(0, nb, 500)):
            f.write(blob)
    big.close()
    assert kmerpos_annot3.main(["--prepare", f"{d}/db", f"{d}/loc"]) == 0
    print(f"{nb} reads, {bases} bases, big.fq.gz {os.path.getsize(f'{d}/big.fq.gz') / 2**30:.2f} GiB", flush=True)
    for threads in [int(x) for x in os.environ.get("CLI_GZ_THREADS", "32,16,24,48,32").split(",")]:
        os.environ["GAVISUNK_GZ_THREADS"] = str(threads)
        t0 = time.perf_counter()
        rc = kmerpos_annot3.main([f"{d}/big.fq.gz", f"{d}/db", f"{d}/loc", f"{d}/out"])
        dt = time.perf_counter() - t0
        print(f"inflate threads {threads}: rc {rc}, {dt:.2f} s = {bases / dt / 1e9:.2f} Gbase/s, phases "
              f"{({k: round(v, 2) for k, v in kmerpos_annot3.last_timing.items()})}", flush=True)
