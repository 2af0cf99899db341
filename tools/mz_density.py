#!/usr/bin/env python3
"""Density (region fetches per window) of the minimiser scheme that addresses the SUNK table, and of two candidates for its t-mer
ORDER that are computable per lane: a decycling-set-first order (Pellow et al. 2023: t-mers of Mykkeltveit's minimum decycling set
sort before all others) and an open-closed-syncmer order (Groot Koerkamp, Liu & Pibiri 2025: t-mers whose smallest s-mer sits in
the middle first, those with it at either end next).  CPU only (numpy), no GPU minutes.

The scheme (csrc/gs_mz.h): a window = one k-mer of the read; among its k - t + 1 t-mers (t = m mod w, w = k - m + 1, raised by
multiples of w) the smallest by ORDER, leftmost on ties, at offset x; the region comes from the canonical m-mer at offset x mod w.
t-mers are ordered as the smaller of themselves and their reverse complement (the probe never asks which strand is canonical).
A RUN = adjacent windows with the same region value: one region fetch.  Reported: runs per window on a random sequence (the
kernel adds 1/64 per window for runs cut at its 64-window step boundaries: 0.2308 -> 0.243 measured).

usage: mz_density.py [n_bases]"""
import sys

import numpy as np


def kmers(seq, n):
    """codes of all n-mers of a uint8 (0..3) sequence, as uint64"""
    out = np.zeros(len(seq) - n + 1, np.uint64)
    for i in range(n):
        out = (out << np.uint64(2)) | seq[i:len(seq) - n + 1 + i].astype(np.uint64)
    return out


def revcomp(code, n):
    out = np.zeros_like(code)
    c = code.copy()
    for _ in range(n):
        out = (out << np.uint64(2)) | (np.uint64(3) - (c & np.uint64(3)))
        c >>= np.uint64(2)
    return out


def mix(x):
    x = (x ^ (x >> np.uint64(31))) * np.uint64(0x7FB5D329728EA185)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x81DADEF4BC2DD44D)
    return x ^ (x >> np.uint64(33))


def decycling_member(code, n):
    """Mykkeltveit's minimum decycling set, by the embedding of Pellow et al.: x in D iff Im(sum s_j w^j) > 0 and the same sum of x
    rotated right by one is <= 0 (w = exp(2 pi i / n)); the all-equal necklaces (sum 0) are put in by their own rule -- ignored here
    (4 t-mers of 4^t)."""
    j = np.arange(n)
    s = np.sin(2 * np.pi * j / n)
    digits = np.stack([((code >> np.uint64(2 * (n - 1 - i))) & np.uint64(3)).astype(np.float64) for i in range(n)], 1)
    im = digits @ s
    im_rot = np.roll(digits, 1, axis=1) @ s
    eps = 1e-9
    return (im > eps) & (im_rot <= eps)


def syncmer_class(code, n, s):
    """0 = open syncmer (smallest s-mer of the t-mer in the middle), 1 = closed (at either end), 2 = neither"""
    ns = n - s + 1
    sm = np.stack([mix((code >> np.uint64(2 * (n - s - i))) & np.uint64((1 << (2 * s)) - 1)) for i in range(ns)], 1)
    a = sm.argmin(1)
    mid = (ns - 1) // 2
    return np.where(a == mid, 0, np.where((a == 0) | (a == ns - 1), 1, 2)).astype(np.uint64)


def density(seq, k, m, order_of, t=None):
    w = k - m + 1
    if t is None:
        t = m % w
        while t < 7:
            t += w
    tm = kmers(seq, t)
    tc = np.minimum(tm, revcomp(tm, t))
    order = order_of(tc, t)
    nt = k - t + 1
    nwin = len(seq) - k + 1
    # sliding leftmost minimum over nt consecutive t-mers: order made unique by position (leftmost wins)
    key = (order << np.uint64(24)) >> np.uint64(24)  # keep 40 bits of order
    best = None
    bestpos = None
    for x in range(nt):
        cand = key[x:x + nwin]
        if best is None:
            best, bestpos = cand.copy(), np.zeros(nwin, np.int64)
        else:
            better = cand < best
            best = np.where(better, cand, best)
            bestpos = np.where(better, x, bestpos)
    p = bestpos % w
    mm = kmers(seq, m)
    mc = np.minimum(mm, revcomp(mm, m))
    region = mc[np.arange(nwin) + p]
    runs = 1 + int((region[1:] != region[:-1]).sum())
    return runs / nwin, t


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    rng = np.random.default_rng(7)
    seq = rng.integers(0, 4, n, dtype=np.uint8)
    orders = {
        "random order on the canonical t-mer (the kernel's)": lambda c, t: mix(c),
        "decycling set first, then random": lambda c, t: (np.where(decycling_member(c, t), np.uint64(0), np.uint64(1)) << np.uint64(39)) | (mix(c) >> np.uint64(25)),
    }
    for s in (3, 4, 5):
        orders[f"open-closed syncmer classes (s = {s}), then random"] = (lambda s_: lambda c, t: (syncmer_class(c, t, s_) << np.uint64(38)) | (mix(c) >> np.uint64(26)))(s)
    print(f"{n} random bases; runs (= region fetches) per window; forward-scheme lower bound ceil((w + m) / w) / (w + m)")
    for k, m in ((20, 15), (21, 15), (20, 14), (25, 19)):
        w = k - m + 1
        bound = -(-(w + m) // w) / (w + m)
        print(f"\nk = {k}, m = {m}, w = {w}: bound {bound:.4f}, random minimiser 2 / (w + 1) = {2 / (w + 1):.4f}")
        for name, fn in orders.items():
            ts = sorted({tt for tt in range(m % w, m + 1, w) if tt >= 5})
            row = []
            for t in ts:
                try:
                    d, _ = density(seq, k, m, fn, t)
                    row.append(f"t = {t}: {d:.4f}")
                except Exception as e:  # (a syncmer needs s < t)
                    row.append(f"t = {t}: -")
            print(f"  {name:58s} " + "   ".join(row))


if __name__ == "__main__":
    main()
