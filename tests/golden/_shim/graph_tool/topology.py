"""See graph_tool/__init__.py in this directory: test-infrastructure stand-in."""
import numpy as np


def _labels(g):
    parent = list(range(g.n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in g.edges:
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    lab, nxt = {}, 0
    out = np.zeros(g.n, dtype=np.int64)
    for v in range(g.n):  # components numbered in order of their lowest vertex
        r = find(v)
        if r not in lab:
            lab[r] = nxt
            nxt += 1
        out[v] = lab[r]
    return out


def label_components(g):
    from . import _VProp

    lab = _labels(g)
    return _VProp(lab), np.bincount(lab) if g.n else np.zeros(0, dtype=np.int64)


def extract_largest_component(g, directed=None, prune=False):
    from . import GraphView

    lab = _labels(g)
    if g.n == 0:
        return GraphView(g, vfilt=np.zeros(0, dtype=bool))
    best = int(np.bincount(lab).argmax())  # first maximum on ties
    return GraphView(g, vfilt=lab == best)
