#!/usr/bin/env python3
"""Lap times of the on-device record parser on ONE genome-scale text block (~1.2 GB of four-line FASTQ, quals '#': what the CLI
stages per block of a `.fq.gz`): GAVISUNK_INGEST_TIMING prints the laps of gs_reads_from_staged_text to stderr.  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import engine  # noqa: E402

rng = np.random.default_rng(1)
n, ln = 20000, 30000
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * ln, dtype=np.uint8)].tobytes()
qual = b"#" * ln
text = b"".join(b"@r%d\n" % i + bases[i * ln:(i + 1) * ln] + b"\n+\n" + qual + b"\n" for i in range(n))
print(f"text {len(text) / 1e9:.2f} GB, {n} reads", flush=True)
os.environ["GAVISUNK_INGEST_TIMING"] = "1"
for rep in range(3):
    t0 = time.perf_counter()
    st = engine.StagedText(text, 0)
    t1 = time.perf_counter()
    dev = engine.DeviceReads.from_staged(st, 20)
    t2 = time.perf_counter()
    st.close()
    dev.close()
    t3 = time.perf_counter()
    print(f"rep {rep}: stage {t1 - t0:.3f} s ({len(text) / (t1 - t0) / 1e9:.1f} GB/s), parse {t2 - t1:.3f} s ({len(text) / (t2 - t1) / 1e9:.1f} GB/s), close {t3 - t2:.3f} s", flush=True)
