#!/usr/bin/env python3
"""Genome-scale surrogate for the HBM-resident (generic) table regime of BASELINE.json configs[2..4]:
a random genome of G Mb (nearly every 20-mer is a SUNK), reads sampled from it with ONT-like errors.
Reports table size, kernel time and bases/s, and checks a prefix of the reads against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gavisunk_amd import synth, engine, _lib

G = int(sys.argv[1]) if len(sys.argv) > 1 else 100          # Mb
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
k = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 20
rng = np.random.default_rng(30)
t0 = time.perf_counter()
contigs = [rng.integers(0, 4, G * 1_000_000 // 4, dtype=np.uint8) for _ in range(4)]
codes, cid, pos = [], [], []
for ci, c in enumerate(contigs):
    cc = synth.canonical_codes(c, k)
    codes.append(cc); cid.append(np.full(len(cc), ci, np.uint32)); pos.append(np.arange(len(cc), dtype=np.int64))
codes = np.concatenate(codes); cid = np.concatenate(cid); pos = np.concatenate(pos)
_, first, counts = np.unique(codes, return_index=True, return_counts=True)
keep = np.sort(first[counts == 1])
codes, cid, pos = codes[keep], cid[keep], pos[keep]
new = np.ones(len(pos), bool); new[1:] = (cid[1:] != cid[:-1]) | ((pos[1:] - pos[:-1]) > k)
loc = (np.cumsum(new) - 1).astype(np.uint32)
if "--many-groups" in sys.argv:  # genome-like group counts: a new group every ~200 SUNKs
    loc = (np.arange(len(pos)) // 200).astype(np.uint32)
print(f"genome {G} Mb: {len(codes)} SUNKs in {int(loc[-1]) + 1} groups ({time.perf_counter() - t0:.0f} s)", flush=True)
lens = synth.lognormal_lengths(rng, n_reads, median=19000.0)
reads = [synth.codes_to_ascii(r) for r in synth.simulate_reads(rng, contigs, lens)]
bases = sum(map(len, reads))
t0 = time.perf_counter()
tab = engine.DeviceTable(k, codes, loc, np.arange(len(codes), dtype=np.uint32))
print(f"table: {tab.layout} | {tab.nbytes / 2**30:.2f} GiB, built in {time.perf_counter() - t0:.1f} s", flush=True)
dev = engine.DeviceReads(engine.ReadBatch.from_ascii(reads), k)
L = _lib.lib()
for _ in range(2):
    st, hits, carry, ctr = engine.tag(tab, dev)
L.gs_counters_reset(tab._h)
for _ in range(5):
    st, hits, carry, ctr = engine.tag(tab, dev)
ms = ctr["probe_ms"] / ctr["launches"]
h_raw = ctr["raw_hits"] / ctr["windows"]
print(f"{n_reads} reads, {bases} bases, h_raw {h_raw:.3f}, {len(hits)} rows: probe {ms:.2f} ms -> {bases / ms / 1e6:.1f} Gbase/s, "
      f"algorithmic {bases * (8.25 + 16 * h_raw) / ms / 1e6:.0f} GB/s = {bases * (8.25 + 16 * h_raw) / ms / 1e6 / 8000:.3f} of the HBM roofline", flush=True)
if "--check" in sys.argv:
    from oracle import oracle
    t0 = time.perf_counter()
    ot = oracle.Table(k, codes, codes, np.zeros(len(codes), np.uint32), loc.astype(np.int64))
    n = 200
    buf, off = oracle.concat_reads(reads[:n])
    rc, oh, _, octr = ot.tag(buf, off)
    d2 = engine.DeviceReads(engine.ReadBatch.from_ascii(reads[:n]), k)
    st, gh, _, gctr = engine.tag(tab, d2)
    ok = len(gh) == len(oh) and np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"]) \
        and np.array_equal(loc[gh["payload"]], loc[oh["row"]]) and gctr["raw_hits"] == octr["raw_hits"]
    print(f"oracle check on {n} reads ({len(oh)} rows): {'bit-exact' if ok else 'MISMATCH'} ({time.perf_counter() - t0:.0f} s)")
