#!/usr/bin/env python3
"""Randomised cross-check of the small per-read GPU kernels against the Python oracle (run on the GPU box; not part of
the test suite): diag_filter_v3 + diag_filter_step2, process-by-contig (pair / component / regions), the bad-SUNK caller,
and the GPU SUNK definition against its NumPy restatement.
usage: fuzz_filters.py [seconds] [first seed]
"""
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import badsunks, diag_filter_step2, diag_filter_v3, engine, process_by_contig, synth  # noqa: E402
from oracle import oracle, sunkdef  # noqa: E402


def fuzz_diag(seed, tmp):
    rng = np.random.default_rng(seed)
    ncontig = int(rng.integers(1, 70))
    contigs = [f"c{i}_{int(rng.integers(0, 10**6))}" for i in range(ncontig)] + ["notinfai1", "notinfai2"]
    in_fai = contigs[:ncontig]
    rows = []
    for r in range(int(rng.integers(1, 400))):
        # (gs_diag.hip takes a read by its row count: <= 128 / 256 / 512 / 1024 rows in LDS, longer ones in global scratch)
        n = int(rng.integers(1, int(rng.choice([3, 12, 60, 200, 3, 12, 60, 200, 3, 12, 60, 200, 130, 300, 700, 1500, 3000]))))
        picks = rng.choice(len(contigs), size=int(rng.integers(1, min(len(contigs), 48) + 1)), replace=False)
        base = int(rng.integers(0, 10**6))
        pos = 0
        sign = 1 if rng.random() < 0.5 else -1
        for _ in range(n):
            pos += int(rng.integers(0, 3000))
            c = contigs[int(rng.choice(picks))]
            u = rng.random()
            jitter = int(rng.integers(-3000, 3000)) if u < 0.3 else (int(rng.choice([-2500, -2499, 2499, 2500])) if u < 0.35 else int(rng.integers(-50, 50)))
            start = base + sign * pos + jitter + (10**7 if sign < 0 else 0)
            rows.append((f"r{r}", pos, c, start, (start // int(rng.choice([1, 500, 5000]))) * 500))
    text = "".join(f"{a}\t{b}\t{c}\t{d}\t{e}\n" for a, b, c, d, e in rows)
    p, fai = os.path.join(tmp, "x.sunkpos"), os.path.join(tmp, "x.fai")
    open(p, "w").write(text)
    open(fai, "w").write("".join(f"{c}\t1\n" for c in in_fai))
    buf = io.StringIO()
    assert diag_filter_v3.run(p, fai, out=buf) == 0
    exp = oracle.diag_filter_v3(oracle.parse_sunkpos(text), in_fai)
    if buf.getvalue() != exp:
        return f"diag_filter_v3 differs ({buf.getvalue().count(chr(10))} / {exp.count(chr(10))} lines)"
    d1 = os.path.join(tmp, "x.diag")
    open(d1, "w").write(exp)
    buf2 = io.StringIO()
    assert diag_filter_step2.run(p, d1, out=buf2) == 0
    exp2 = oracle.diag_filter_step2(text, exp)
    if buf2.getvalue() != exp2:
        return "diag_filter_step2 differs"
    return None


def fuzz_pair(seed, tmp):
    rng = np.random.default_rng(seed)
    rows, rlen = [], []
    span = int(rng.choice([10**5, 4 * 10**6]))
    for r in range(int(rng.integers(1, 200))):
        n = int(rng.integers(1, int(rng.choice([4, 30, 120]))))
        L = int(rng.choice([900, 5000, 20000, 80000]))
        rlen.append(f"read{r:03d}\t{L}")
        base = int(rng.integers(0, span))
        orient = 1 if rng.random() < 0.5 else -1
        p = 0
        seen = []
        for i in range(n):
            p += int(rng.integers(0, 2000))
            true_start = base + orient * p + (3 * span if orient < 0 else 0)
            u = rng.random()
            if u < 0.1:
                s = int(rng.integers(0, 4 * span))
            elif u < 0.2 and seen:
                s = seen[int(rng.integers(0, len(seen)))]
            else:
                s = true_start + int(rng.integers(-40, 40))
            seen.append(s)
            rows.append((f"read{r:03d}", p, "ctgX", s, (s // 3000) * 3000))
    rng.shuffle(rows)
    rows.sort(key=lambda t: (int(t[0][4:]) % 7, t[1]))
    text = "".join("\t".join(map(str, t)) + "\n" for t in rows)
    rl = "\n".join(rlen) + "\n"
    groups = sorted({t[4] for t in rows})
    bad = "".join(f"ctgX:{g}\n" for g in groups if rng.random() < 0.05)
    sp, rlp, bd = (os.path.join(tmp, x) for x in ("c.sunkpos", "c.rlen", "c.bad"))
    open(sp, "w").write(text), open(rlp, "w").write(rl), open(bd, "w").write(bad)
    tsv, bed = os.path.join(tmp, "o.tsv"), os.path.join(tmp, "o.bed")
    for f in (tsv, bed):
        if os.path.exists(f):
            os.remove(f)
    assert process_by_contig.main(["unused.loc", sp, rlp, bd, tsv, bed]) == 0
    etsv, ebed = oracle.process_by_contig(text, rl, bad)
    if open(tsv).read() != etsv:
        return "process_by_contig tsv differs"
    if (open(bed).read() if os.path.exists(bed) else "") != ebed:
        return "process_by_contig bed differs"
    return None


def fuzz_sunkdef(seed, tmp):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([16, 20, 20, 21, 25, 31, 32]))
    fam = rng.integers(0, 4, int(rng.integers(30, 900)), dtype=np.uint8)
    contigs = []
    for i in range(int(rng.integers(1, 9))):
        n = int(rng.choice([0, 1, k - 1, k, k + 1, 500, 12000, 60000]))
        c = rng.integers(0, 4, n, dtype=np.uint8)
        for _ in range(int(rng.integers(0, 1 + n // 4000))):
            q = int(rng.integers(0, max(n - len(fam), 1)))
            c[q:q + len(fam)] = fam[: len(c[q:q + len(fam)])]
        a = bytearray(synth.codes_to_ascii(c))
        for q in rng.integers(0, max(n, 1), n // int(rng.choice([500, 3000, 10**9]))):
            a[int(q)] = ord("N")
        if rng.random() < 0.3:
            a = bytearray(bytes(a).lower())
        contigs.append((f"c{i}", bytes(a)))
    if contigs[0][1] and rng.random() < 0.5:  # a reverse-complement copy: those k-mers stop being unique
        src = synth.ascii_to_codes(contigs[0][1][:4000].upper().replace(b"N", b"A"))
        contigs.append(("rc0", synth.codes_to_ascii(synth.revcomp_codes(src))))
    exp = sunkdef.define_sunks(contigs, k)
    got = engine.define_sunks(contigs, k)
    same = (len(got) == len(exp) and np.array_equal(got.contig_id, exp.contig_id) and np.array_equal(got.start, exp.start)
            and np.array_equal(got.code, exp.code) and np.array_equal(got.group, exp.group))
    return None if same else f"define_sunks differs at k {k} ({len(got)} / {len(exp)} SUNKs)"


def fuzz_badsunks(seed, tmp):
    rng = np.random.default_rng(seed)
    h1 = [f"h1_{i}" for i in range(int(rng.integers(1, 5)))]
    h2 = [f"h2_{i}" for i in range(int(rng.integers(1, 5)))]

    def sp(own, other):
        rows = []
        for r in range(int(rng.integers(1, 500))):
            for _ in range(int(rng.integers(1, 30))):
                c = own[int(rng.integers(0, len(own)))] if rng.random() < 0.8 else other[int(rng.integers(0, len(other)))]
                g = int(rng.zipf(1.4)) % 400 * 1000
                rows.append(f"r{r}\t{int(rng.integers(0, 50000))}\t{c}\t{g + 5}\t{g}\n")
        return "".join(rows)

    fai1 = "".join(f"{c}\t10\n" for c in h1)
    fai2 = "".join(f"{c}\t10\n" for c in h2)
    sp1, sp2 = sp(h1, h2), sp(h2, h1)
    paths = {n: os.path.join(tmp, n) for n in ("f1", "f2", "s1", "s2", "out")}
    for n, t in (("f1", fai1), ("f2", fai2), ("s1", sp1), ("s2", sp2)):
        open(paths[n], "w").write(t)
    assert badsunks.main([paths["f1"], paths["f2"], paths["s1"], paths["s2"], paths["out"]]) == 0
    got = set(open(paths["out"]).read().split())
    return None if got == oracle.badsunks(fai1, fai2, sp1, sp2) else "badsunks differs"


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, rounds = time.time(), 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() - t0 < budget:
            for name, fn in (("diag", fuzz_diag), ("pair", fuzz_pair), ("sunkdef", fuzz_sunkdef), ("badsunks", fuzz_badsunks)):
                err = fn(seed, tmp)
                if err:
                    print(f"MISMATCH {name} seed {seed}: {err}", flush=True)
                    return 1
            rounds += 1
            seed += 1
            if rounds % 20 == 0:
                print(f"{rounds} rounds, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {rounds} rounds (seeds up to {seed - 1})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
