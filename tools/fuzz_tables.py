#!/usr/bin/env python3
"""Randomised cross-check of the two table-file parsers (gs_sunkdb_load: `<kmers.db>`, `<kmer.loc>`) and of the host join.  CPU only.
Every round writes a k-mer list and a location table with the irregularities real files and broken ones have -- CRLF, empty lines,
a last line without newline, lower case and N bases, k-mers listed twice, loc rows that repeat a k-mer or carry one that is not in
the list, contigs that come back, negative and large numbers, and (one round in five) a malformed row -- and loads them with the
multi-threaded parsers (several thread counts) and with the single-threaded line readers: identical arrays, contig order, location
ids and, for a malformed file, the identical error text.  The join is checked against a dictionary in Python.
usage: fuzz_tables.py [seconds] [first seed]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gavisunk_amd import _lib, engine  # noqa: E402

ENC = {c: v for cs, v in (("Aa", 0), ("Cc", 1), ("Gg", 2), ("TtUu", 3)) for c in cs}  # nim-kmer's encode: anything else -> A


def code(s: bytes) -> int:
    c = 0
    for ch in s.decode("latin1"):
        c = (c << 2) | ENC.get(ch, 0)
    return c & 0xFFFFFFFFFFFFFFFF


def make(rng, k, big):
    n = int(rng.integers(120_000, 200_000)) if big else int(rng.integers(0, 400))
    lut = np.frombuffer(b"ACGT", np.uint8)
    words = [w.tobytes() for w in lut[rng.integers(0, 4, (max(n, 1), k), dtype=np.uint8)]]
    eol = b"\r\n" if rng.random() < 0.2 else b"\n"

    def spell(w):
        u = rng.random()
        if u < 0.03:
            return w.lower()
        if u < 0.05:
            i = int(rng.integers(0, k))
            return w[:i] + b"N" + w[i + 1:]
        return w

    db = [spell(words[int(i)]) for i in rng.integers(0, len(words), n)] if n else []
    rows = []
    contigs = [b"ctg%d" % i for i in range(int(rng.integers(1, 40)))] + [b"weird name|x", b"c"]
    c = 0
    start = 0
    for j in range(int(n * rng.uniform(0.5, 1.5))):
        if rng.random() < 0.001:
            c = int(rng.integers(0, len(contigs)))
        start = start + int(rng.integers(0, 50)) if rng.random() > 0.002 else int(rng.integers(-5, 10**10))
        w = spell(words[int(rng.integers(0, len(words)))]) if rng.random() < 0.95 else lut[rng.integers(0, 4, k, dtype=np.uint8)].tobytes()
        rows.append(b"%s\t%d\t%s\t%d" % (contigs[c], start, w, (start // 1000) * 1000) + (b"\textra" if rng.random() < 0.01 else b""))
        if rng.random() < 0.002:
            rows.append(b"")
    bad_at = None
    if rows and rng.random() < 0.2:
        bad_at = int(rng.integers(0, len(rows)))
        rows[bad_at] = [b"only\ttwo", b"c\tx12\tACGT\t5", b"c\t12\tACGT\t", b"no tabs at all", b"c\t1\tACGT\t1e3"][int(rng.integers(0, 5))]
    tail = b"" if rng.random() < 0.3 else eol
    return eol.join(db) + (tail if db else b""), eol.join(rows) + (tail if rows else b""), bad_at


def load(dbp, locp, threads):
    if threads is None:
        os.environ.pop("GAVISUNK_LOC_THREADS", None)
    else:
        os.environ["GAVISUNK_LOC_THREADS"] = str(threads)
    try:
        db = engine.SunkDB(dbp, locp)
    except _lib.GsError as e:
        return ("error", e.status, e.msg)
    out = {k_: np.array(getattr(db, k_)) for k_ in ("row_contig", "row_start", "row_group", "row_loc", "keys", "key_row", "key_loc")}
    out["contigs"], out["k"] = list(db.contigs), db.k
    db.close()
    return out


def one_round(seed, tmp):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([11, 20, 20, 25, 32]))
    big = rng.random() < 0.5
    dbt, loct, bad_at = make(rng, k, big)
    dbp, locp = os.path.join(tmp, "db"), os.path.join(tmp, "loc")
    open(dbp, "wb").write(dbt)
    open(locp, "wb").write(loct)
    ref = load(dbp, locp, 1)
    for th in (None, 2, 5):
        got = load(dbp, locp, th)
        if isinstance(ref, tuple) or isinstance(got, tuple):
            if ref != got:
                return f"seed {seed}: threads {th}: {got if isinstance(got, tuple) else 'loaded'} against {ref if isinstance(ref, tuple) else 'loaded'}"
            continue
        if got["contigs"] != ref["contigs"] or got["k"] != ref["k"]:
            return f"seed {seed}: threads {th}: contigs / k differ"
        for k_ in ref:
            if k_ not in ("contigs", "k") and not np.array_equal(got[k_], ref[k_]):
                return f"seed {seed}: threads {th}: {k_} differs"
    if isinstance(ref, tuple) and not dbt.strip() and "k = 0" in ref[2]:
        return None  # (an empty k-mer list has no k: refused by every path alike, checked above)
    if (bad_at is not None) != isinstance(ref, tuple):
        # (a malformed row in a file whose k-mer list is empty is still an error; an extra fifth column is not)
        return f"seed {seed}: malformed row at {bad_at}: {'loaded' if not isinstance(ref, tuple) else ref}"
    if isinstance(ref, tuple) or big:
        return None
    # the join against a dictionary (small files): distinct db codes in first-seen order, the LAST loc row of each
    db_lines = dbt.replace(b"\r\n", b"\n").split(b"\n")
    if db_lines and db_lines[-1] == b"" and dbt.endswith((b"\n",)):
        db_lines.pop()
    keys, seen = [], set()
    for ln in db_lines:
        c = code(ln)
        if c not in seen:
            seen.add(c)
            keys.append(c)
    last = {}
    r = 0
    for ln in loct.replace(b"\r\n", b"\n").split(b"\n"):
        if not ln:
            continue
        last[code(ln.split(b"\t")[2])] = r
        r += 1
    want_row = [last.get(c, _lib.GS_MISSING) for c in keys]
    if ref["keys"].tolist() != keys or ref["key_row"].tolist() != want_row:
        return f"seed {seed}: the host join differs from the dictionary's ({len(keys)} keys)"
    return None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, rounds = time.time(), 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() - t0 < budget:
            err = one_round(seed, tmp)
            if err:
                print("MISMATCH", err, flush=True)
                return 1
            rounds += 1
            seed += 1
            if rounds % 20 == 0:
                print(f"{rounds} rounds, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {rounds} rounds (seeds up to {seed - 1})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
