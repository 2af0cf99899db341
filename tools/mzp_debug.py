"""Debug helper for the place-record table (csrc/gs_mzp.h): small random cases through the exact per-lane form (build + global
lookups) and through the fast kernel, against the oracle; prints the first differing row.  GAVISUNK_LIB selects the library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gavisunk_amd import _lib, engine, synth  # noqa: E402
from oracle import oracle, sunkdef  # noqa: E402


def case(seed, k, glen, n_reads, median, div, err=0.01):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, glen, dtype=np.uint8)
    g[glen // 2:glen // 2 + 800] = g[1000:1800]
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], k)
    loc = (sunks.start // div).astype(np.uint32)
    lens = synth.lognormal_lengths(rng, n_reads, median=median, sigma=0.8, lo=1, hi=glen - 1)
    seqs = [synth.codes_to_ascii(r) for r in synth.simulate_reads(rng, [g], lens, sub=err, dele=err, ins=err)]
    return sunks.code, loc, seqs


def run(layout, k, keys, loc, seqs):
    os.environ["GAVISUNK_TABLE"] = layout
    tab = engine.DeviceTable(k, keys, loc, np.arange(len(keys), dtype=np.uint32))
    name = tab.layout
    dev = engine.DeviceReads(engine.ReadBatch.from_ascii(seqs), k)
    st, hits, carry, ctr = engine.tag(tab, dev)
    dev.close()
    tab.close()
    return st, hits, carry, ctr, name


def main():
    bad = 0
    for seed, k, glen, n_reads, median, div, err in [(1, 20, 30000, 30, 1500, 50, 0.01), (2, 20, 200000, 100, 4000, 400, 0.02),
                                                     (3, 20, 60000, 40, 3000, 1, 0.0), (4, 21, 50000, 40, 2000, 50, 0.01),
                                                     (5, 16, 50000, 40, 2000, 50, 0.01), (6, 20, 1_500_000, 200, 8000, 2000, 0.02)]:
        keys, loc, seqs = case(seed, k, glen, n_reads, median, div, err)
        t = oracle.Table(k, keys, keys, np.zeros(len(keys), np.uint32), loc.astype(np.int64))
        buf, off = oracle.concat_reads(seqs)
        rc, oh, oprev, octr = t.tag(buf, off, cap=1 << 22)
        for layout in ("minimizer-exact", "minimizer"):
            st, gh, carry, ctr, name = run(layout, k, keys, loc, seqs)
            ok = (st == 0 and ctr["raw_hits"] == octr["raw_hits"] and len(gh) == len(oh) and np.array_equal(gh["read"], oh["read"])
                  and np.array_equal(gh["pos"], oh["pos"]) and np.array_equal(gh["payload"], oh["row"]))
            print(f"seed {seed} k {k} {layout:16s} {'ok ' if ok else 'BAD'} raw {ctr['raw_hits']} / {octr['raw_hits']} rows {len(gh)} / {len(oh)} fallback {ctr['fallback_tiles']} | {name[:150]}")
            if not ok:
                bad += 1
                n = min(len(gh), len(oh))
                d = np.nonzero((gh["read"][:n] != oh["read"][:n]) | (gh["pos"][:n] != oh["pos"][:n]) | (gh["payload"][:n] != oh["row"][:n]))[0]
                if len(d):
                    i = int(d[0])
                    print("   first difference at row", i, "gpu", gh[max(i - 1, 0):i + 2].tolist(), "oracle", oh[max(i - 1, 0):i + 2].tolist())
    print("bad:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
