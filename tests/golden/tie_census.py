#!/usr/bin/env python3
"""Tie census of the inter-SUNK stage's goldens (tests/golden/pair_*).

Those goldens are outputs of the reference's own process-by-contig_lowmem_AR.py run under pandas 2.x with a graph_tool
STAND-IN (make_golden_pair.py): everything up to the graph calls is the reference's code executing for real, but two
decisions are the stand-in's / the installed pandas', not the pinned environment's:

  (1) `value_counts().index[0]` when two positions of one ID are equally frequent (process-by-contig_lowmem_AR.py:171) --
      the order among equal counts is pandas-version dependent;
  (2) `extract_largest_component` when two components have the same, largest size (:195) -- the tie rule is graph-tool's.

This script recomputes, read by read, the quantities those decisions depend on (from the file-level filtering of
oracle.per_read_inputs and a few lines of its own: pair mask, orientation, surviving pairs) and counts how often a tie
occurs.  A golden with zero ties of both kinds cannot contain a stand-in decision: whatever rule the pinned pandas /
graph-tool apply, it never had to choose.  (What stays assumed even then: the ORDER of a read's IDs in the TSV, i.e.
that `add_edge_list(hashed=True)` numbers vertices by first appearance -- graph-tool's documented behaviour.)

    python tests/golden/tie_census.py            # table over all pair goldens, exit 1 if any tie exists
"""
from __future__ import annotations

import itertools
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = ["pair_amy", "pair_multi_k20", "pair_multi_k25", "pair_synth", "pair_long"]


def census_read(ids, pos, start):
    """-> dict(pairs, kept, multipos_ids, value_count_ties, components, largest_ties, orientation_tie)."""
    n = len(ids)
    pairs = list(itertools.combinations(range(n), 2))
    kept = []
    for i, j in pairs:
        sd, pd_ = abs(float(start[i]) - float(start[j])), abs(float(pos[i]) - float(pos[j]))
        ratio = (pd_ / sd) if sd else (float("inf") if pd_ else float("nan"))
        if not (ratio >= 1.1 or ratio <= 0.9):
            kept.append((i, j, 1 if pos[i] > pos[j] else 0))
    out = dict(pairs=len(pairs), kept=len(kept), multipos_ids=0, value_count_ties=0, components=0, largest_ties=0, orientation_tie=0)
    if not kept:
        return out
    ones = sum(s for _, _, s in kept)
    zeros = len(kept) - ones
    out["orientation_tie"] = int(ones == zeros)  # numpy's argmax over [0, 1]: deterministic (0), not a stand-in matter
    orient = 1 if ones > zeros else 0
    edges = [(i, j) for i, j, s in kept if s == orient]
    counts = {}
    for i, j in edges:
        for r in (i, j):
            counts.setdefault(ids[r], {}).setdefault(pos[r], 0)
    for i, _ in edges:
        counts[ids[i]][pos[i]] += 1
    for _, j in edges:
        counts[ids[j]][pos[j]] += 1
    drop = set()
    for idv, c in counts.items():
        if len(c) < 2:
            continue
        out["multipos_ids"] += 1
        top = max(c.values())
        winners = [p for p, v in c.items() if v == top]
        if len(winners) > 1:
            out["value_count_ties"] += 1
        good = winners[0]
        for e, (i, j) in enumerate(edges):
            if ids[i] == idv and not (pos[i] == good or pos[j] == good):
                drop.add(e)
    parent = {}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for e, (i, j) in enumerate(edges):
        if e in drop:
            continue
        for v in (ids[i], ids[j]):
            parent.setdefault(v, v)
        a, b = find(ids[i]), find(ids[j])
        if a != b:
            parent[a] = b
    sizes = {}
    for v in parent:
        sizes[find(v)] = sizes.get(find(v), 0) + 1
    out["components"] = len(sizes)
    if sizes:
        top = max(sizes.values())
        out["largest_ties"] = int(sum(1 for v in sizes.values() if v == top) > 1)
    return out


def census_case(case: str):
    from oracle import oracle

    d = os.path.join(HERE, case)
    meta = json.load(open(os.path.join(d, "case.json")))
    tot = dict(reads=0, reads_with_pairs=0, multipos_ids=0, value_count_ties=0, largest_ties=0, orientation_ties=0, max_hits=0)
    for c in meta["contigs"]:
        t = c["tag"]
        _, groups = oracle.per_read_inputs(open(f"{d}/{t}.sunkpos").read(), open(f"{d}/rlen").read(), open(f"{d}/{t}.bad").read())
        for rname, g in (groups or {}).items():
            r = census_read([x[4] for x in g], [x[1] for x in g], [x[3] for x in g])
            tot["reads"] += 1
            tot["reads_with_pairs"] += int(r["kept"] > 0)
            tot["multipos_ids"] += r["multipos_ids"]
            tot["value_count_ties"] += r["value_count_ties"]
            tot["largest_ties"] += r["largest_ties"]
            tot["orientation_ties"] += r["orientation_tie"]
            tot["max_hits"] = max(tot["max_hits"], len(g))
    return tot


def main():
    bad = 0
    for case in CASES:
        t = census_case(case)
        print(f"{case:16s} reads {t['reads']:5d}  with kept pairs {t['reads_with_pairs']:5d}  max hits/read {t['max_hits']:4d}  "
              f"multi-position IDs {t['multipos_ids']:4d}  value_counts ties {t['value_count_ties']:3d}  "
              f"largest-component ties {t['largest_ties']:3d}  (orientation ties {t['orientation_ties']})")
        bad += t["value_count_ties"] + t["largest_ties"]
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
