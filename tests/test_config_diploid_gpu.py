"""BASELINE.json configs[3] ("config 4"): a DIPLOID assembly -- two 3.0 Gb haplotypes that descend from
one ancestor and differ by 0.1 % SNPs plus 5 % haplotype-specific sequence -- whose SUNKs are the 20-mers singly unique over
BOTH haplotypes (workflow/rules/defineSUNKs.smk:1-61: `combine_asm_haps`, then `jellyfish count -C -U 1`), and ONE GPU's
shard (1/8: ~15 Gbase) of the ~40x ONT read set, drawn from both haplotypes.  Unlike config 3 the table (~3e8 keys) is
almost only heterozygous-site k-mers in millions of small location groups, and most windows miss.

6e9 windows are more than one sort holds (and more than 2^32): gs_sunks_define_device counts them in passes over the code
space.  What is checked, and why that way: tests/_bigcfg.py."""
import numpy as np
import pytest

from _bigcfg import chain_slice_check, full_size_checks
from gavisunk_amd import _lib, engine, synth_gpu

pytestmark = pytest.mark.gpu

K = 20
W = synth_gpu.WORKLOADS["diploid"]


def test_diploid_shard_full_size():
    gen = synth_gpu.make_genome(W)
    assert gen.n_bases > 5.9e9 and len(gen.off) == 49
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, K)
    # the haplotypes differ where they should: a SUNK sits on a SNP or in haplotype-specific sequence, on either haplotype
    pos = sunks.pos(0, 1 << 20)
    assert int(pos[0] >> np.uint64(32)) == 0 and sunks.n > 100_000_000 and sunks.n_groups > (1 << 21)
    table = sunks.table()
    rs = synth_gpu.make_reads(W, gen)
    contig_names = list(gen.names)
    gen.close()
    reads = engine.device_reads_from_device_ascii(rs.ptr, rs.off, K)
    try:
        assert table.layout.startswith("minimizer-hbm") and rs.n_bases > 1.4e10
        n = W["reads"]
        whole, carry, c = full_size_checks(table, reads, rs, sunks, K, cuts=[0, 1, 100_000, 100_001, 366_000, n],
                                           slices=((0, 1500), (90_000, 90_500), (180_000, 181_000), (270_000, 270_500), (n - 800, n)), min_rows=1_000_000,
                                           min_slice_rows=1000, hit_rate=(0.005, 0.2))
        # rows from both haplotypes (contigs 0..23 = hap 1, 24..47 = hap 2)
        L = _lib.lib()
        ctg = synth_gpu.gather(0, L.gs_sunkset_dev_pos(sunks._h), np.uint64, whole["payload"][: 1 << 20]) >> np.uint64(32)
        assert (ctg < 24).any() and (ctg >= 24).any()
        # the device-resident second half over the shard's whole hit list (48 contigs, 3e6 location groups); its head against the oracle
        kept, tsv_rows, regions = chain_slice_check(table, reads, rs, sunks, contig_names, 400, K)
        assert kept > 1_000_000 and tsv_rows > 10_000
    finally:
        reads.close()
        table.close()
        sunks.close()
        rs.close()


def test_definition_in_passes_equals_one_pass(monkeypatch):
    """The multi-pass definition (code space cut into P partitions) gives the same set, rows and groups as one pass."""
    gen = synth_gpu.DiploidGenome(20_000_000, seed=41)
    a = engine.DeviceSunkSet(gen.ptr, gen.off, K)
    monkeypatch.setenv("GAVISUNK_DEFINE_PASSES", "8")
    b = engine.DeviceSunkSet(gen.ptr, gen.off, K)
    gen.close()
    try:
        assert a.n == b.n > 1_000_000 and a.n_groups == b.n_groups
        assert np.array_equal(a.code(), b.code()) and np.array_equal(a.pos(), b.pos())
        assert np.array_equal(a.group(), b.group()) and np.array_equal(a.loc(), b.loc())
    finally:
        a.close()
        b.close()
