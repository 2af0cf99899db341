#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Runs a reference script IN PLACE (runpy, from /root/reference) under the image's pandas 2.x with the
two pandas-1.3 API behaviours workflow/scripts/badsunks_AR.py relies on restored (its pinned environment is pandas 1.3.4,
workflow/envs/viz.yaml:82-100):
  * Series.value_counts(): the result keeps the Series' NAME and its index has none (2.x: named "count", index named after
    the Series), so that `DataFrame(counts).reset_index().rename(columns={'ID2': 'count', 'index': 'ID2'})`
    (badsunks_AR.py:24-27, :63-66) yields the columns ID2 / count it was written for;
  * Series.str.split(pat, n, expand) takes n positionally (badsunks_AR.py:29, :68; keyword-only in 2.x).
Nothing of the script's arithmetic is touched: the counts, the mode, the limit and the queries are its own lines executing on
real pandas.  usage: pandas13_names.py <script> [args ...]"""
import runpy
import sys

import pandas as pd
from pandas.core.strings.accessor import StringMethods

_value_counts = pd.Series.value_counts
_split = StringMethods.split


def value_counts(self, *a, **k):
    r = _value_counts(self, *a, **k)
    r.name = self.name
    r.index.name = None
    return r


def split(self, pat=None, n=-1, expand=False, **k):
    return _split(self, pat, n=n, expand=expand, **k)


pd.Series.value_counts = value_counts
StringMethods.split = split

if __name__ == "__main__":
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")
