"""TEST INFRASTRUCTURE ONLY -- a minimal stand-in for the parts of graph-tool 2.43 that
workflow/scripts/process-by-contig_lowmem_AR.py touches, used by tests/golden/make_golden_pair.py to run
the reference script in the build container (graph-tool itself is not installed).  Semantics follow
graph-tool's documentation: `add_edge_list(hashed=True)` creates vertices in order of first appearance
(row by row, source then target) and returns the map vertex -> original id; `label_components` numbers
components in order of their lowest vertex; `extract_largest_component` keeps the component with the
largest size, the first such label on ties.  Goldens made through this stand-in are exact wherever no
tie is involved and PROVISIONAL for ties."""
import numpy as np

from . import topology  # noqa: F401


class _VProp:
    def __init__(self, arr):
        self._a = np.asarray(arr)

    def get_array(self):
        return self._a

    @property
    def a(self):
        return self._a


class _VPropDict(dict):
    def __getattr__(self, k):
        return self[k]

    def __setattr__(self, k, v):
        self[k] = v


class Graph:
    def __init__(self, directed=True):
        self.directed = directed
        self.edges = []
        self.n = 0
        self.vp = _VPropDict()

    def add_edge_list(self, values, hashed=False, hash_type="int64_t", eprops=None):
        assert hashed
        ids, index = [], {}
        for row in np.asarray(values):
            pair = []
            for v in (row[0], row[1]):
                v = int(v)
                if v not in index:
                    index[v] = len(ids)
                    ids.append(v)
                pair.append(index[v])
            self.edges.append(tuple(pair))
        self.n = len(ids)
        return _VProp(np.array(ids, dtype=np.int64))

    def get_vertices(self):
        return np.arange(self.n)


class GraphView:
    def __init__(self, g, vfilt=None):
        self.g = g
        self.vp = g.vp
        self._keep = np.nonzero(np.asarray(vfilt))[0] if vfilt is not None else np.arange(g.n)

    def get_vertices(self):
        return self._keep
