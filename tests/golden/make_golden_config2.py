#!/usr/bin/env python3
"""Full-size golden for BASELINE.json configs[1] ("config 2": 5 Mb contig, 10 000 reads, k = 20): the reference's own
prebuilt binaries (/root/reference/workflow/scripts/{kmerpos_annot3,rlen,diag_filter_v3,diag_filter_step2}; the
commands of tagONT.smk:36,73,92,110) run on the exact inputs `gavisunk_amd.synth.config2(oracle.sunkdef.define_sunks, seed=20, n_reads=10000)`
produces -- the inputs of tests/test_fullsize_gpu.py and of bench.py's config-2 leg.

The inputs (74 MiB db, 150 MiB loc, 230 MiB FASTA) are regenerated from the seed and never stored; what is kept under
tests/golden/config2_full/ is what the binaries printed (gzip, mtime 0) plus the md5 of every input and output.
Runs only in the build container (the reference tree does not travel).  ~2 minutes.

    python tests/golden/make_golden_config2.py           # (re)generate
    python tests/golden/make_golden_config2.py --check   # regenerate into a temp dir, compare with the committed files
"""
from __future__ import annotations

import argparse
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from gavisunk_amd import synth  # noqa: E402
from oracle import sunkdef

BIN = "/root/reference/workflow/scripts"
K = 20
CONTIG = "ctg5M"


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def decode_many(codes: np.ndarray, k: int) -> np.ndarray:
    """(n, k) uint8 ASCII matrix of 2-bit codes, first base most significant."""
    out = np.empty((len(codes), k), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    for j in range(k):
        out[:, j] = lut[((codes >> np.uint64(2 * (k - 1 - j))) & np.uint64(3)).astype(np.int64)]
    return out


def write_inputs(d):
    seq, sunks, names, reads = synth.config2(sunkdef.define_sunks, seed=20, n_reads=10000, k=K)
    # db: one canonical k-mer per line, ascending code (jellyfish dump order does not matter to the tagger)
    kmers = decode_many(np.sort(sunks.code), K)
    with open(os.path.join(d, "c2.db"), "wb") as f:
        f.write(np.concatenate([kmers, np.full((len(kmers), 1), 10, np.uint8)], axis=1).tobytes())
    # loc: chrom \t start \t kmer \t group, (contig, start) ascending
    km = decode_many(sunks.code, K)
    with open(os.path.join(d, "c2.loc"), "wb") as f:
        step = 200000
        for i in range(0, len(sunks), step):
            rows = [b"%s\t%d\t%s\t%d\n" % (CONTIG.encode(), int(s), km[i + j].tobytes(), int(g))
                    for j, (s, g) in enumerate(zip(sunks.start[i:i + step], sunks.group[i:i + step]))]
            f.write(b"".join(rows))
    synth.write_fasta(os.path.join(d, "c2_reads.fa"), names, reads)
    with open(os.path.join(d, "c2.fai"), "w") as f:
        f.write(f"{CONTIG}\t{len(seq)}\t7\t{len(seq)}\t{len(seq) + 1}\n")
    return len(sunks), int(sum(len(r) for r in reads))


def run_reference(d):
    p = lambda f: os.path.join(d, f)  # noqa: E731
    t0 = time.time()
    subprocess.run([f"{BIN}/kmerpos_annot3", p("c2_reads.fa"), p("c2.db"), p("c2.loc"), p("c2.sunkpos")], check=True,
                   stderr=subprocess.DEVNULL)
    t_tag = time.time() - t0
    subprocess.run([f"{BIN}/rlen", p("c2_reads.fa"), p("c2.rlen")], check=True)
    with open(p("c2.diag"), "wb") as fh:
        subprocess.run([f"{BIN}/diag_filter_v3", p("c2.sunkpos"), p("c2.fai")], check=True, stdout=fh)
    with open(p("c2.diag2"), "wb") as fh:
        subprocess.run([f"{BIN}/diag_filter_step2", p("c2.sunkpos"), p("c2.diag")], check=True, stdout=fh)
    return t_tag


def gz_copy(src, dst):
    with open(src, "rb") as f, open(dst, "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as g:
        g.write(f.read())


def build(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="gs_c2_", dir="/tmp") as d:
        n_sunks, n_bases = write_inputs(d)
        t_tag = run_reference(d)
        meta = {"k": K, "sunks": n_sunks, "reads": 10000, "bases": n_bases,
                "generator": "gavisunk_amd.synth.config2(sunkdef.define_sunks, seed=20, n_reads=10000, k=20)",
                "reference_tagger_seconds_in_build_container": round(t_tag, 1),
                "md5": {f: md5(os.path.join(d, f)) for f in ("c2.db", "c2.loc", "c2_reads.fa", "c2.fai", "c2.sunkpos", "c2.rlen",
                                                            "c2.diag", "c2.diag2")},
                "rows": {f: sum(1 for _ in open(os.path.join(d, f), "rb")) for f in ("c2.sunkpos", "c2.rlen", "c2.diag", "c2.diag2")}}
        for f in ("c2.sunkpos", "c2.diag", "c2.diag2"):
            gz_copy(os.path.join(d, f), os.path.join(out_dir, f + ".gz"))
    meta.pop("reference_tagger_seconds_in_build_container") if out_dir != os.path.join(HERE, "config2_full") else None
    with open(os.path.join(out_dir, "case.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
        fh.write("\n")
    return meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    dest = os.path.join(HERE, "config2_full")
    if a.check:
        with tempfile.TemporaryDirectory(prefix="gs_c2chk_", dir="/tmp") as t:
            meta = build(t)
            old = json.load(open(os.path.join(dest, "case.json")))
            ok = meta["md5"] == old["md5"] and all(
                open(os.path.join(t, f), "rb").read() == open(os.path.join(dest, f), "rb").read()
                for f in ("c2.sunkpos.gz", "c2.diag.gz", "c2.diag2.gz"))
        print("config-2 full-size golden reproduces" if ok else "MISMATCH")
        sys.exit(0 if ok else 1)
    meta = build(dest)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
