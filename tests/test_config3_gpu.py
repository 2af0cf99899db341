"""BASELINE.json configs[2] ("config 3") at FULL size: CHM13-scale haplotype surrogate (24 contigs, 3.1 Gb), its exact SUNK
set (~1.5e9 20-mers, gs_sunks_define_device) in an HBM-resident minimiser-addressed table, 1 000 000 synthetic ONT reads
(~4e10 bases) -- the workload bench.py's headline line is measured on, generated on the GPU by the same code
(synth_gpu.WORKLOADS["config3"]).  What is checked, and why that way: tests/_bigcfg.py."""
import pytest

from _bigcfg import chain_slice_check, full_size_checks
from gavisunk_amd import engine, synth_gpu

pytestmark = pytest.mark.gpu

K = 20
W = synth_gpu.WORKLOADS["config3"]


def test_config3_full_size():
    gen = synth_gpu.make_genome(W)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, K)
    table = sunks.table()
    rs = synth_gpu.make_reads(W, gen)
    contig_names = list(gen.names)
    gen.close()
    reads = engine.device_reads_from_device_ascii(rs.ptr, rs.off, K)
    try:
        assert table.layout.startswith("minimizer-hbm") and sunks.n > 1_000_000_000 and rs.n_bases > 3.5e10
        n = W["reads"]
        full_size_checks(table, reads, rs, sunks, K, cuts=[0, 1, 300_000, 300_001, 999_000, n],
                         slices=((0, 2500), (125_000, 126_000), (250_000, 251_000), (375_000, 376_000), (500_000, 501_500), (625_000, 626_000),
                                 (750_000, 751_000), (n - 2000, n)), min_rows=1_000_000, min_slice_rows=1000)
        # the second half of the engine over this pass's whole hit list, on the device; its head against the oracle
        kept, tsv_rows, regions = chain_slice_check(table, reads, rs, sunks, contig_names, 400, K)
        assert kept > 10_000_000 and tsv_rows > 100_000 and regions > 1000
    finally:
        reads.close()
        table.close()
        sunks.close()
        rs.close()
