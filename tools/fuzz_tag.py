#!/usr/bin/env python3
"""Randomised cross-check of the tagging paths on one GPU (run on the GPU box; not part of the test suite).

Every round draws a genome, a SUNK table (optionally thinned, regrouped, undersized), and a read set with a random mix of
lengths (empty and shorter-than-k reads, reads of exactly one tile, long reads), error rates and homopolymer inserts; tags it
under every table layout and compares the hit arrays, carries and counters bit for bit; small rounds are also checked
against the CPU oracle.  usage: fuzz_tag.py [seconds] [first seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import _lib, engine, synth  # noqa: E402
from oracle import oracle, sunkdef  # noqa: E402

K = 20  # k of the current round (most rounds: GAVISUNK's default 20)
LAYOUTS = ["auto", "compact-hbm", "minimizer", "minimizer-exact", "minimizer-slots", "generic"]


def draw(seed):
    global K
    rng = np.random.default_rng(seed)
    K = 20 if rng.random() < 0.6 else int(rng.choice([11, 16, 17, 18, 19, 21, 22, 23, 25, 28, 31, 32]))
    glen = int(rng.choice([3000, 40000, 300000, 1500000]))
    g = rng.integers(0, 4, glen, dtype=np.uint8)
    for _ in range(int(rng.integers(0, 6))):  # repeats: their k-mers are not SUNKs
        ln = int(rng.integers(50, max(51, glen // 20)))
        a, b = (int(x) for x in rng.integers(0, glen - ln, 2))
        g[a:a + ln] = g[b:b + ln]
    if rng.random() < 0.3:
        a = int(rng.integers(0, glen - 500))
        g[a:a + 400] = 0  # a poly-A stretch in the assembly
    sunks = sunkdef.define_sunks([("c", synth.codes_to_ascii(g))], K)
    keys, start = sunks.code, sunks.start
    if len(keys) > 10 and rng.random() < 0.4:  # thin the table: lower hit rate
        keep = rng.random(len(keys)) < rng.uniform(0.05, 0.9)
        keys, start = keys[keep], start[keep]
    div = int(rng.choice([1, 7, 50, 400, 5000]))
    loc = (start // div).astype(np.uint32)
    if rng.random() < 0.3 and len(loc):
        loc = (loc % int(rng.integers(1, 9))).astype(np.uint32)  # few groups, revisited
    order = np.argsort(keys, kind="stable")
    keys, loc = keys[order].astype(np.uint64), loc[order]
    n_reads = int(rng.choice([1, 5, 60, 400, 2500]))
    median = float(rng.choice([30, 500, 2100, 9000, 40000]))
    lens = synth.lognormal_lengths(rng, n_reads, median=median, sigma=float(rng.uniform(0.2, 1.2)), lo=0, hi=glen - 1)
    err = float(rng.choice([0.0, 0.003, 0.02]))
    reads = synth.simulate_reads(rng, [g], np.maximum(lens, 1), sub=err, dele=err, ins=err)
    seqs = []
    for i, r in enumerate(reads):
        s = synth.codes_to_ascii(r)[: int(lens[i])]
        x = rng.random()
        if x < 0.03:
            s = b""
        elif x < 0.06:
            s = s[: int(rng.integers(0, K))]
        elif x < 0.09:
            s = s[: K + 2047]  # exactly one tile of windows (when the read is that long)
        elif x < 0.12:
            s = s + b"A" * int(rng.integers(1, 30000))
        elif x < 0.14:
            s = b"T" * int(rng.integers(1, 3000)) + s
        elif x < 0.16:
            s = s.replace(b"A", b"N", 3).lower()
        seqs.append(s)
    env = {}
    if rng.random() < 0.25:
        env["GAVISUNK_COMPACT_QB"] = str(int(rng.integers(4, 13)))
    return keys, loc, seqs, env


def tag(layout, keys, loc, seqs, cuts):
    if layout == "auto":
        os.environ.pop("GAVISUNK_TABLE", None)
    else:
        os.environ["GAVISUNK_TABLE"] = layout
    tab = engine.DeviceTable(K, keys, loc, np.arange(len(keys), dtype=np.uint32))
    name = tab.layout.split(":")[0]
    carry, parts, raw, win = _lib.GS_NONE, [], 0, 0
    for a, b in zip(cuts, cuts[1:]):
        batch = engine.ReadBatch.from_ascii(seqs[a:b])
        dev = engine.DeviceReads(batch, K)
        batch.close()
        st, h, carry, ctr = engine.tag(tab, dev, carry)
        assert st == 0
        h = h.copy()
        h["read"] += a
        parts.append(h)
        raw += ctr["raw_hits"]
        win += ctr["windows"]
        dev.close()
    tab.close()
    return np.concatenate(parts), carry, raw, win, name


def one_round(seed, hot=None, verbose=False):
    """One seeded round: every layout against the first one, small rounds against the oracle too.  hot: GAVISUNK_MZ_HOT for the
    round (None: the library's own rule; 0: never a hot region; n: regions with n misfits become hot).
    -> (ok, message, {layout name: 1})"""
    keys, loc, seqs, env = draw(seed)
    for k in ("GAVISUNK_COMPACT_QB", "GAVISUNK_MZ_HOT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    if hot is not None:
        os.environ["GAVISUNK_MZ_HOT"] = str(hot)
    names = {}
    try:
        rng = np.random.default_rng(seed + 7)
        n = len(seqs)
        cuts = [0, n] if rng.random() < 0.5 else sorted({0, n, *(int(x) for x in rng.integers(0, n + 1, 3))})
        ref = None
        for layout in LAYOUTS:
            if verbose:  # which (seed, layout) a GPU fault belongs to
                print(f"seed {seed} k {K} layout {layout} keys {len(keys)} reads {len(seqs)} env {env} cuts {cuts}", flush=True)
            got = tag(layout, keys, loc, seqs, cuts if layout != "generic" else [0, n])
            names[got[4]] = names.get(got[4], 0) + 1
            if ref is None:
                ref = got
            else:
                ok = np.array_equal(got[0], ref[0]) and got[1:4] == ref[1:4]
                if not ok:
                    return False, (f"MISMATCH seed {seed} k {K} layout {layout} ({got[4]}) vs {ref[4]}: rows {len(got[0])} / {len(ref[0])}, "
                                   f"carry {got[1]} / {ref[1]}, raw {got[2]} / {ref[2]}, env {env}, hot {hot}, cuts {cuts}"), names
        if ref[3] < 3_000_000:  # the oracle too
            t = oracle.Table(K, keys, keys, np.zeros(len(keys), np.uint32), loc.astype(np.int64))
            buf, off = oracle.concat_reads(seqs)
            rc, oh, oprev, octr = t.tag(buf, off)
            ok = (rc == 0 and len(oh) == len(ref[0]) and np.array_equal(oh["read"], ref[0]["read"])
                  and np.array_equal(oh["pos"], ref[0]["pos"]) and np.array_equal(oh["row"], ref[0]["payload"])
                  and octr["raw_hits"] == ref[2] and octr["windows"] == ref[3])
            if not ok:
                return False, f"ORACLE MISMATCH seed {seed} k {K}: rows {len(oh)} / {len(ref[0])}, env {env}, hot {hot}", names
        return True, "", names
    finally:
        for k in ("GAVISUNK_COMPACT_QB", "GAVISUNK_MZ_HOT", "GAVISUNK_TABLE"):
            os.environ.pop(k, None)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, rounds, names = time.time(), 0, {}
    while time.time() - t0 < budget:
        ok, msg, seen = one_round(seed, hot=(None, 0, 3)[seed % 3], verbose=bool(os.environ.get("FUZZ_VERBOSE")))
        for k_, v in seen.items():
            names[k_] = names.get(k_, 0) + v
        if not ok:
            print(msg, flush=True)
            return 1
        rounds += 1
        seed += 1
        if rounds % 10 == 0:
            print(f"{rounds} rounds, {time.time() - t0:.0f} s, layouts seen {names}", flush=True)
    print(f"OK: {rounds} rounds (seeds up to {seed - 1}), layouts seen {names}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
