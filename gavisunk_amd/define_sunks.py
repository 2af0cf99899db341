"""GPU SUNK definition: the inputs of the tagger straight from the assemblies.

    python -m gavisunk_amd.define_sunks -k 20 hap1.fa[.gz] hap2.fa[.gz] --db jellyfish.db --loc kmer.loc

replaces, for this path's purposes, rules jellyfish_count / define_SUNKs / mrsfast_index / mrsfast_search /
bed_convert of workflow/rules/defineSUNKs.smk (:22-127): `--db` gets one canonical k-mer per line (ascending code;
jellyfish's own dump order is its hash order and carries no meaning), `--loc` gets `chrom start kmer group` rows in
(contig, start) order, the two files `kmerpos_annot3` reads."""
from __future__ import annotations

import argparse
import os
import sys

from . import engine, synth


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("-k", type=int, default=20)
    ap.add_argument("assemblies", nargs="+")
    ap.add_argument("--db", required=True)
    ap.add_argument("--loc", required=True)
    a = ap.parse_args(argv)
    contigs = []
    for path in a.assemblies:
        contigs += synth.read_fasta(path)
    s = engine.define_sunks(contigs, a.k, device=int(os.environ.get("GAVISUNK_DEVICE", "0")))
    s.write_db(a.db)
    s.write_loc(a.loc)
    sys.stderr.write(f"{len(s)} SUNKs (k={a.k}) over {len(contigs)} contigs\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
