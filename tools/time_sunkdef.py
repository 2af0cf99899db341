#!/usr/bin/env python3
"""Timing of the GPU SUNK definition (N1) against the NumPy restatement at config-2 and larger sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gavisunk_amd import synth, engine
from oracle import sunkdef
for mb in (5, 100):
    if mb == 5:
        genome, _ = synth.genome_config2(20)
    else:
        genome = np.random.default_rng(1).integers(0, 4, mb * 1_000_000, dtype=np.uint8)
    contigs = [("ctg", synth.codes_to_ascii(genome))]
    engine.define_sunks(contigs[:1], 20)  # warm-up (HIP init, rocPRIM kernels)
    t0 = time.perf_counter(); g = engine.define_sunks(contigs, 20); tg = time.perf_counter() - t0
    line = f"{mb} Mb: GPU {tg:.3f} s ({len(g)} SUNKs, {len(np.unique(g.group))} groups)"
    if mb == 5:
        t0 = time.perf_counter(); e = sunkdef.define_sunks(contigs, 20); te = time.perf_counter() - t0
        ok = np.array_equal(g.code, e.code) and np.array_equal(g.start, e.start) and np.array_equal(g.group, e.group)
        line += f", NumPy {te:.2f} s, identical={ok}"
    print(line, flush=True)
