#!/usr/bin/env python3
"""Seconds by phase of the command-line drop-in at 1e8 keys WITHOUT the sidecar (db and loc parsed, joined and built on every run,
as the reference does per scatter chunk): the files bench.py's `cli_1e8_keys` leg writes, three runs.  GPU box."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import engine, kmerpos_annot3, synth_gpu  # noqa: E402

os.environ["GAVISUNK_TIMING"] = "1"
with tempfile.TemporaryDirectory(prefix="gs_cli1e8_", dir="/tmp") as d:
    gen = synth_gpu.Genome(200_000_000, n_contigs=24, seed=31, device=0)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, 20, 0)
    sunks.write_files(gen.names, f"{d}/db", f"{d}/loc")
    rng = np.random.default_rng(31)
    rs = synth_gpu.ReadSet(gen, synth_gpu.lognormal_lengths(rng, 4000, median=32000.0), seed=31000)
    gen.close()
    with open(f"{d}/reads.fa", "wb") as f:
        for i, sq in enumerate(rs.host_reads(0, rs.n_reads)):
            f.write(b">r%d\n" % i + sq + b"\n")
    print(f"{sunks.n} keys, {rs.n_bases} bases; db {os.path.getsize(f'{d}/db') / 2**30:.1f} GiB, loc {os.path.getsize(f'{d}/loc') / 2**30:.1f} GiB", flush=True)
    rs.close()
    sunks.close()
    for rep in range(3):
        t0 = time.perf_counter()
        rc = kmerpos_annot3.main([f"{d}/reads.fa", f"{d}/db", f"{d}/loc", f"{d}/out"])
        print(f"run {rep}: rc {rc}, {time.perf_counter() - t0:.2f} s, phases {({k: round(v, 3) for k, v in kmerpos_annot3.last_timing.items()})}", flush=True)
