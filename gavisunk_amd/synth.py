"""Synthetic inputs for the SUNK-tagging path: assemblies and ONT-like reads.  BENCH / TEST SUPPORT, not on the product hot path.

It exists because the reference ships no read data (`.test/data/*.ONT.fa.gz` are absent) and because the
benchmark configurations of BASELINE.json are defined on synthetic surrogates.  The SUNK set of
an assembly comes from a `define` callable the caller supplies: the product's GPU definition (engine.define_sunks) in bench.py,
the oracle's NumPy restatement of jellyfish / mrsfast / bedtools (oracle/sunkdef.py) in the tests -- the restatement itself is
checker code and lives with the oracle, not here.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import numpy as np

_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.zeros(256, dtype=np.uint8)  # every byte that is not C/G/T/U maps to 0 ('A')
for _ch, _v in ((b"Cc", 1), (b"Gg", 2), (b"TtUu", 3)):
    for _b in _ch:
        _CODE[_b] = _v
_IS_ACGT = np.zeros(256, dtype=bool)
for _b in b"ACGTacgt":
    _IS_ACGT[_b] = True


def ascii_to_codes(seq: bytes | np.ndarray) -> np.ndarray:
    a = np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else seq
    return _CODE[a]


def codes_to_ascii(codes: np.ndarray) -> bytes:
    return _ASCII[codes].tobytes()


def revcomp_codes(codes: np.ndarray) -> np.ndarray:
    return (3 - codes[::-1]).astype(np.uint8)


def canonical_codes(codes: np.ndarray, k: int) -> np.ndarray:
    """Canonical 2-bit k-mer code (min of forward and reverse complement, first base
    most significant) for every window of `codes`; length len-k+1 (empty if len<k)."""
    n = len(codes) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    fwd = np.zeros(n, dtype=np.uint64)
    rc = np.zeros(n, dtype=np.uint64)
    c64 = codes.astype(np.uint64)
    for j in range(k):
        w = c64[j:j + n]
        fwd = (fwd << np.uint64(2)) | w
        rc |= (np.uint64(3) - w) << np.uint64(2 * j)
    return np.minimum(fwd, rc)


def decode_kmer(code: int, k: int) -> str:
    return "".join("ACGT"[(code >> (2 * (k - 1 - j))) & 3] for j in range(k))


@dataclass
class SunkSet:
    """SUNKs of an assembly.  Arrays are in `kmer.loc` row order: (contig, start) ascending."""
    k: int
    contig_names: List[str]
    contig_id: np.ndarray  # u32 per SUNK
    start: np.ndarray      # i64 per SUNK, 0-based
    code: np.ndarray       # u64 canonical code per SUNK
    group: np.ndarray      # i64 per SUNK (start of merged interval)

    def __len__(self) -> int:
        return len(self.code)

    def db_lines(self) -> Iterable[str]:
        """`jellyfish dump -c`-like list: one canonical k-mer per line (ascending code)."""
        for c in np.sort(self.code):
            yield decode_kmer(int(c), self.k)

    def write_db(self, path: str) -> None:
        with open(path, "w") as f:
            for s in self.db_lines():
                f.write(s + "\n")

    def write_loc(self, path: str) -> None:
        with open(path, "w") as f:
            for ci, st, cd, g in zip(self.contig_id, self.start, self.code, self.group):
                f.write(f"{self.contig_names[ci]}\t{int(st)}\t{decode_kmer(int(cd), self.k)}\t{int(g)}\n")


# ----------------------------------------------------------------------------------
# genomes
# ----------------------------------------------------------------------------------

def genome_config2(seed: int = 20, length: int = 5_000_000, n_families: int = 5,
                   family_len: int = 3000, n_copies: int = 500,
                   rng: np.random.Generator | None = None) -> Tuple[np.ndarray, np.random.Generator]:
    """SURVEY.md Appendix B genome: uniform ACGT with pasted repeat-family copies."""
    rng = rng or np.random.default_rng(seed)
    seq = rng.integers(0, 4, length, dtype=np.uint8)
    fams = [rng.integers(0, 4, family_len).astype(np.uint8) for _ in range(n_families)]
    for c in range(n_copies):
        p = int(rng.integers(0, length - family_len))
        seq[p:p + family_len] = fams[c % n_families]
    return seq, rng


def read_fasta(path: str) -> List[Tuple[str, bytes]]:
    op = gzip.open if path.endswith(".gz") else open
    out, name, chunks = [], None, []
    with op(path, "rb") as f:
        for line in f:
            line = line.rstrip()
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, b"".join(chunks)))
                name, chunks = line[1:].split()[0].decode(), []
            else:
                chunks.append(line)
    if name is not None:
        out.append((name, b"".join(chunks)))
    return out


# ----------------------------------------------------------------------------------
# reads
# ----------------------------------------------------------------------------------

def lognormal_lengths(rng: np.random.Generator, n: int, median: float = 20000.0, sigma: float = 0.7,
                      lo: int = 1000, hi: int = 400000) -> np.ndarray:
    return np.clip(rng.lognormal(np.log(median), sigma, n).astype(np.int64), lo, hi)


def mutate(rng: np.random.Generator, codes: np.ndarray, sub: float, dele: float, ins: float) -> np.ndarray:
    """ONT-like errors: substitution / deletion / duplicated-base insertion per position."""
    n = len(codes)
    x = rng.random(n)
    out = codes.copy()
    s = x < sub
    out[s] = rng.integers(0, 4, int(s.sum()), dtype=np.uint8)
    rep = np.ones(n, dtype=np.int64)
    rep[(x >= sub) & (x < sub + dele)] = 0
    rep[(x >= sub + dele) & (x < sub + dele + ins)] = 2
    return np.repeat(out, rep)


def simulate_reads(rng: np.random.Generator, contigs: Sequence[np.ndarray], lengths: Sequence[int],
                   sub: float = 0.02, dele: float = 0.02, ins: float = 0.02,
                   p_reverse: float = 0.5) -> List[np.ndarray]:
    """Reads (2-bit code arrays) drawn uniformly from `contigs` (code arrays)."""
    sizes = np.array([len(c) for c in contigs], dtype=np.float64)
    reads = []
    for ln in lengths:
        ci = int(rng.choice(len(contigs), p=sizes / sizes.sum())) if len(contigs) > 1 else 0
        c = contigs[ci]
        ln = int(min(ln, len(c) - 1))
        st = int(rng.integers(0, len(c) - ln))
        r = c[st:st + ln]
        if rng.random() < p_reverse:
            r = revcomp_codes(r)
        reads.append(mutate(rng, r, sub, dele, ins))
    return reads


def write_fasta(path: str, names: Sequence[str], reads: Sequence[np.ndarray], width: int = 0,
                fastq: bool = False) -> None:
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wb") as f:
        for n, r in zip(names, reads):
            s = r if isinstance(r, (bytes, bytearray)) else codes_to_ascii(r)
            if fastq:
                f.write(b"@" + n.encode() + b"\n" + s + b"\n+\n" + b"#" * len(s) + b"\n")
            elif width:
                f.write(b">" + n.encode() + b"\n")
                for i in range(0, len(s), width):
                    f.write(s[i:i + width] + b"\n")
            else:
                f.write(b">" + n.encode() + b"\n" + s + b"\n")


def config2(define, seed: int = 20, n_reads: int = 10000, k: int = 20, median: float = 20000.0):
    """BASELINE.json configs[1]: 5 Mb contig, 10 k reads (N50 ~ 30 kb), k = 20.  `define(contigs, k) -> SunkSet`: the SUNK
    definition to use (engine.define_sunks on the GPU, oracle.sunkdef.define_sunks as the checker)."""
    seq, rng = genome_config2(seed)
    sunks = define([("ctg5M", codes_to_ascii(seq))], k)
    lens = lognormal_lengths(rng, n_reads, median)
    reads = simulate_reads(rng, [seq], lens)
    names = [f"r{i}" for i in range(n_reads)]
    return seq, sunks, names, reads
