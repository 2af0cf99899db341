"""Synthetic genome-scale inputs generated ON the GPU (libgssynth.so, csrc/gs_synth.hip).  BENCH / TEST SUPPORT, not product.

BASELINE.json configs[2] ("config 3"): a CHM13-scale haplotype -- 24 contigs, 3.1 Gb, about half of it
in repeat-family copies -- and 1 000 000 ONT-like reads (lognormal lengths, N50 ~ 50 kb, 6 % errors, both strands).  Real
CHM13 / ONT data do not exist offline; like config 2 this is a size- and density-matched surrogate, and its SUNK set is
the EXACT one (gs_sunks_define_device over the generated assembly).  At 3.1 Gb nothing here may touch the host: the
assembly, its 10^9-key table and the 4 * 10^10 read bases are produced, packed and tagged in HBM.

No torch: a torch wheel bundles its own HIP runtime, and two HIP runtimes cannot both hold the GPU in one process.
Device buffers are plain device pointers (ints) owned by the objects below.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgssynth.so")
_lib = None
P, U32, U64, I, D = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m gavisunk_amd.build`")
        L = C.CDLL(LIB_PATH)
        L.gs_synth_last_error.restype = C.c_char_p
        L.gs_synth_free.argtypes = [I, P]
        L.gs_synth_free.restype = None
        L.gs_synth_genome.argtypes = [I, U64, U64, D, U32, U32, U32, C.POINTER(P)]
        L.gs_synth_diploid.argtypes = [I, U64, U64, D, U32, U32, D, D, U32, C.POINTER(P)]
        L.gs_synth_reads.argtypes = [I, U64, P, U32, P, P, P, D, D, D, C.POINTER(P), P]
        L.gs_synth_select_rows.argtypes = [I, P, U64, P, U64, P, U64, C.POINTER(U64)]
        L.gs_synth_gather.argtypes = [I, P, I, P, U64, P]
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc:
        raise RuntimeError(f"libgssynth: {lib().gs_synth_last_error().decode(errors='replace')} (status {rc})")


def _to_host(device: int, ptr: int, nbytes: int) -> bytes:
    from . import _lib as gl

    buf = np.empty(nbytes, dtype=np.uint8)
    if nbytes:
        gl.check(gl.lib().gs_copy_to_host(device, buf.ctypes.data, ptr, nbytes))
    return buf.tobytes()


class Genome:
    """ASCII assembly in device memory: `n_contigs` contigs back to back, sizes falling off linearly (like chromosomes
    1..22, X, Y roughly); slots of `family_len` bases are exact repeat-family copies with probability `repeat_fraction`."""

    def __init__(self, total_bases: int, n_contigs: int = 24, repeat_fraction: float = 0.5, n_families: int = 64,
                 family_len: int = 6000, seed: int = 30, device: int = 0, break_period: int = 0):
        """break_period > 0: a 44-base interspersed element every break_period bases of unique sequence, so that the runs of SUNKs
        -- the location groups -- are about that long instead of as long as the unique slots (12 000 on average)."""
        wts = np.linspace(2.0, 0.5, n_contigs)
        lens = np.maximum((wts / wts.sum() * total_bases).astype(np.int64), 1000)
        self.off = np.zeros(n_contigs + 1, dtype=np.uint64)
        self.off[1:] = np.cumsum(lens)
        self.names = [f"ctg{i + 1}" for i in range(n_contigs)]
        self.device = device
        p = P()
        _check(lib().gs_synth_genome(device, seed, int(self.off[-1]), repeat_fraction, n_families, family_len, break_period, C.byref(p)))
        self.ptr = p.value

    @property
    def n_bases(self) -> int:
        return int(self.off[-1])

    def host(self, lo: int, hi: int) -> bytes:
        return _to_host(self.device, self.ptr + lo, hi - lo)

    def close(self):
        if getattr(self, "ptr", None):
            lib().gs_synth_free(self.device, self.ptr)
            self.ptr = None

    __del__ = close


class DiploidGenome(Genome):
    """BASELINE.json configs[3] (SURVEY.md section 8d cfg 4): two haplotypes of `bases_per_hap` that descend from one ancestor
    (the assembly `Genome` makes from the same arguments) and differ by `snp` = 0.1 % SNPs plus `spec` = 5 % haplotype-specific
    sequence (slots of `spec_len` bases replaced, per haplotype, by sequence found nowhere else).  Hap 1's contigs, then hap
    2's, back to back in one device buffer -- what `combine_asm_haps` (workflow/rules/defineSUNKs.smk:1-20) hands the SUNK
    definition; a k-mer is a SUNK when it is singly unique over BOTH haplotypes."""

    def __init__(self, bases_per_hap: int, n_contigs: int = 24, repeat_fraction: float = 0.5, n_families: int = 64,
                 family_len: int = 6000, snp: float = 0.001, spec: float = 0.05, spec_len: int = 20000, seed: int = 40,
                 device: int = 0):
        wts = np.linspace(2.0, 0.5, n_contigs)
        lens = np.maximum((wts / wts.sum() * bases_per_hap).astype(np.int64), 1000)
        per_hap = int(lens.sum())
        self.off = np.zeros(2 * n_contigs + 1, dtype=np.uint64)
        self.off[1:] = np.cumsum(np.concatenate([lens, lens]))
        self.names = [f"hap{h}_ctg{i + 1}" for h in (1, 2) for i in range(n_contigs)]
        self.device = device
        self.bases_per_hap = per_hap
        p = P()
        _check(lib().gs_synth_diploid(device, seed, per_hap, repeat_fraction, n_families, family_len, snp, spec, spec_len, C.byref(p)))
        self.ptr = p.value


def lognormal_lengths(rng: np.random.Generator, n: int, median: float, sigma: float = 0.7, lo: int = 1000,
                      hi: int = 400000) -> np.ndarray:
    return np.clip(rng.lognormal(np.log(median), sigma, n).astype(np.int64), lo, hi)


class ReadSet:
    """Reads drawn uniformly from the contigs (contig chosen by size), mutated like synth.mutate (2 % substitutions, 2 %
    deletions, 2 % insertions), half of them from the reverse strand; ASCII, back to back, in device memory."""

    def __init__(self, gen: Genome, lengths: np.ndarray, seed: int, sub: float = 0.02, dele: float = 0.02, ins: float = 0.02,
                 p_reverse: float = 0.5):
        rng = np.random.default_rng(seed)
        n = len(lengths)
        clen = (gen.off[1:] - gen.off[:-1]).astype(np.int64)
        ci = rng.choice(len(clen), size=n, p=clen / clen.sum())
        ln = np.minimum(np.asarray(lengths, dtype=np.int64), clen[ci] - 1)
        st = gen.off[ci].astype(np.int64) + (rng.random(n) * (clen[ci] - ln)).astype(np.int64)
        rev = (rng.random(n) < p_reverse).astype(np.uint8)
        st = np.ascontiguousarray(st, np.uint64)
        ln32 = np.ascontiguousarray(ln, np.uint32)
        self.off = np.zeros(n + 1, dtype=np.uint64)
        self.device = gen.device
        p = P()
        _check(lib().gs_synth_reads(gen.device, seed, gen.ptr, n, st.ctypes.data, ln32.ctypes.data, rev.ctypes.data, sub, dele, ins,
                                    C.byref(p), self.off.ctypes.data))
        self.ptr = p.value
        self.src_start, self.src_len, self.rev = st, ln32, rev

    @property
    def n_reads(self) -> int:
        return len(self.off) - 1

    @property
    def n_bases(self) -> int:
        return int(self.off[-1])

    def host_reads(self, lo: int, hi: int) -> List[bytes]:
        a, b = int(self.off[lo]), int(self.off[hi])
        buf = _to_host(self.device, self.ptr + a, b - a)
        return [buf[int(self.off[i]) - a:int(self.off[i + 1]) - a] for i in range(lo, hi)]

    def close(self):
        if getattr(self, "ptr", None):
            lib().gs_synth_free(self.device, self.ptr)
            self.ptr = None

    __del__ = close


def select_rows(device: int, d_codes: int, n: int, queries: np.ndarray) -> np.ndarray:
    """Sorted indices i with d_codes[i] (u64, device) in `queries` (host)."""
    q = np.unique(np.ascontiguousarray(queries, np.uint64))
    rows = np.empty(max(len(q), 1), dtype=np.uint32)
    cnt = U64(0)
    _check(lib().gs_synth_select_rows(device, d_codes, n, q.ctypes.data, len(q), rows.ctypes.data, len(rows), C.byref(cnt)))
    return np.sort(rows[: cnt.value])


def gather(device: int, d_src: int, dtype, rows: np.ndarray) -> np.ndarray:
    rows = np.ascontiguousarray(rows, np.uint32)
    out = np.empty(len(rows), dtype=dtype)
    _check(lib().gs_synth_gather(device, d_src, out.itemsize, rows.ctypes.data, len(rows), out.ctypes.data))
    return out


def n50(lengths: np.ndarray) -> int:
    s = np.sort(np.asarray(lengths, dtype=np.int64))[::-1]
    return int(s[np.searchsorted(np.cumsum(s), s.sum() / 2)])


# ------------------------------------------------------------------------------------------------------------------
# The BASELINE.json workloads, defined ONCE for bench.py and the full-size GPU tests
# ------------------------------------------------------------------------------------------------------------------
WORKLOADS = {
    # configs[2] ("config 3"): CHM13-scale haplotype, 1 000 000 reads N50 ~ 50 kb, k = 20
    "config3": dict(kind="haploid", genome_mb=3100.0, repeat_fraction=0.5, reads=1_000_000, median=32000.0, max_len=400_000, seed=30,
                    title="config 3 (BASELINE.json configs[2]): CHM13-scale haplotype surrogate"),
    # configs[3] ("config 4"): diploid assembly (both haplotypes), ~40x ONT = 1.2e11 bases read-sharded over 8 GPUs: ONE shard
    "diploid": dict(kind="diploid", genome_mb=3000.0, repeat_fraction=0.5, snp=0.001, spec=0.05, spec_len=20000, reads=367_000,
                    median=32000.0, max_len=400_000, seed=40,
                    title="config 4 (BASELINE.json configs[3]): diploid assembly surrogate (2 x 3.0 Gb, 0.1 % SNPs + 5 % hap-specific), "
                          "one GPU's shard (1/8) of ~40x ONT"),
    # configs[4] ("config 5"): T2T primate assembly, 60 % repeats, ultra-long reads N50 = 100 kb, k = 21 and k = 25
    "config5": dict(kind="haploid", genome_mb=3000.0, repeat_fraction=0.6, reads=500_000, median=61300.0, max_len=1_000_000, seed=50,
                    title="config 5 (BASELINE.json configs[4]): T2T primate assembly surrogate (60 % repeats), ultra-long ONT"),
}


def make_genome(w: dict, genome_mb: float | None = None, device: int = 0) -> Genome:
    mb = w["genome_mb"] if genome_mb is None else genome_mb
    if w["kind"] == "diploid":
        return DiploidGenome(int(mb * 1e6), n_contigs=24, repeat_fraction=w["repeat_fraction"], snp=w["snp"], spec=w["spec"],
                             spec_len=w["spec_len"], seed=w["seed"], device=device)
    return Genome(int(mb * 1e6), n_contigs=24, repeat_fraction=w["repeat_fraction"], seed=w["seed"], device=device,
                  break_period=int(w.get("group_len", 0)))


def make_reads(w: dict, gen: Genome, n_reads: int | None = None, shard: int = 0) -> ReadSet:
    """Shard `shard` of the workload's read set (every shard its own seeded draw: at N GPUs rank r tags shard r)."""
    n = w["reads"] if n_reads is None else n_reads
    rng = np.random.default_rng([w["seed"], shard])
    return ReadSet(gen, lognormal_lengths(rng, n, median=w["median"], hi=w["max_len"]), seed=w["seed"] * 1000 + shard)
