"""CPU restatement of GAVISUNK's SUNK definition.  TEST INFRASTRUCTURE: the checker of
gs_sunks_define / gs_sunks_define_device and the source of the fixtures' SUNK sets; nothing under gavisunk_amd/ imports it.

What `workflow/rules/defineSUNKs.smk` obtains from third-party tools that are not in the build image (jellyfish 2.3.0, mrsfast
3.4.2, bedtools 2.30: workflow/envs/viz.yaml:82-100), restated with NumPy from the rules' command lines:

* `jellyfish count -m {SUNK_len} -C ... -U 1` + `jellyfish dump -c -U 1` (defineSUNKs.smk:39,59): canonical k-mers (-C: a
  k-mer and its reverse complement count as one) whose count over ALL contigs of both haplotypes (`combine_asm_haps`, :1-20)
  is exactly 1; windows with a non-ACGT byte are not k-mers;
* `mrsfast --search ... -e 0` (:101): their exact positions -- with count 1, the one place the k-mer (either strand) occurs;
* `bedtools bamtobed | bedtools merge` (:124-125, default -d 0): overlapping AND book-ended `[start, start + k)` intervals
  merge, i.e. a new group starts where the gap between consecutive SUNK starts on a contig exceeds k;
* `bedtools intersect -wa -wb` (:126): every SUNK row gets the start of its merged interval as group id; rows in
  (contig, start) order.

PARITY UNPINNED: the reference holds no fixture for this stage and the tools cannot run here; tests/test_sunkdef_gpu.py adds a
second, independent brute-force counter (a Python dict over tiny inputs) so that the GPU path is not checked against one
restatement alone.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from gavisunk_amd.synth import _CODE, _IS_ACGT, SunkSet, canonical_codes


def define_sunks(contigs: Sequence[Tuple[str, bytes]], k: int) -> SunkSet:
    """Exact SUNK set of `contigs` [(name, ascii sequence)] (both haplotypes together)."""
    codes_all, cid_all, pos_all = [], [], []
    for ci, (_, seq) in enumerate(contigs):
        a = np.frombuffer(seq, dtype=np.uint8)
        cods = _CODE[a]
        canon = canonical_codes(cods, k)
        n = len(canon)
        if n == 0:
            continue
        # windows containing a non-ACGT byte are not k-mers (jellyfish skips them)
        bad = np.concatenate(([0], np.cumsum(~_IS_ACGT[a])))
        ok = (bad[k:k + n] - bad[:n]) == 0
        idx = np.nonzero(ok)[0]
        codes_all.append(canon[idx])
        cid_all.append(np.full(len(idx), ci, dtype=np.uint32))
        pos_all.append(idx.astype(np.int64))
    codes = np.concatenate(codes_all) if codes_all else np.zeros(0, np.uint64)
    cid = np.concatenate(cid_all) if cid_all else np.zeros(0, np.uint32)
    pos = np.concatenate(pos_all) if pos_all else np.zeros(0, np.int64)
    _, first, counts = np.unique(codes, return_index=True, return_counts=True)
    keep = np.sort(first[counts == 1])  # original order = (contig, pos) ascending
    codes, cid, pos = codes[keep], cid[keep], pos[keep]
    group = pos.copy()
    if len(pos):
        new = np.ones(len(pos), dtype=bool)
        new[1:] = (cid[1:] != cid[:-1]) | ((pos[1:] - pos[:-1]) > k)
        head = np.maximum.accumulate(np.where(new, np.arange(len(pos)), 0))  # index of each run's first SUNK
        group = pos[head]
    return SunkSet(k, [n for n, _ in contigs], cid, pos, codes, group.astype(np.int64))
