"""BASELINE.json configs[4] ("config 5"): SUNK lengths other than 20 -- k = 21 and k = 25 (the `SUNK_len`
knob, config/config.yaml:2; kmerpos_annot3.nim:28-38 takes k from the db) -- at the configuration's own size: a 3 Gb
assembly surrogate with 60 % of it in repeat families, its exact SUNK set (~1e9 keys; one table at a time fits HBM), and
500 000 ultra-long synthetic ONT reads (N50 ~ 100 kb, ~3.9e10 bases), generated on the GPU by the same code as bench.py's
(synth_gpu.WORKLOADS["config5"]).  What is checked, and why that way: tests/_bigcfg.py."""
import pytest

from _bigcfg import full_size_checks
from gavisunk_amd import engine, synth_gpu

pytestmark = pytest.mark.gpu

W = synth_gpu.WORKLOADS["config5"]


@pytest.mark.parametrize("k", [21, 25])
def test_other_sunk_lengths_full_size(k):
    gen = synth_gpu.make_genome(W)
    sunks = engine.DeviceSunkSet(gen.ptr, gen.off, k)
    table = sunks.table()
    rs = synth_gpu.make_reads(W, gen)
    gen.close()
    reads = engine.device_reads_from_device_ascii(rs.ptr, rs.off, k)
    try:
        # place records under a minimiser of k - 6 bases (w = 7: a tag word holds the 2 x 6 flank bases): k = 21: m = 15, one
        # register per minimiser value (k_probe_mz<21,15,.,1>); k = 25: m = 19, 38 bits in two (k_probe_mz<25,19,.,2>)
        assert table.layout.startswith("minimizer-hbm") and f"mod-minimiser of length {k - 6} (w = 7" in table.layout
        assert "place records" in table.layout
        assert sunks.n > 800_000_000 and rs.n_bases > 3.5e10 and synth_gpu.n50(rs.off[1:] - rs.off[:-1]) > 95_000
        n = W["reads"]
        full_size_checks(table, reads, rs, sunks, k, cuts=[0, 1, 150_000, 150_001, 499_000, n],
                         slices=((0, 1500), (125_000, 126_000), (250_000, 251_000), (375_000, 376_000), (n - 1000, n)), min_rows=1_000_000, min_slice_rows=300,
                         hit_rate=(0.05, 0.4))
    finally:
        reads.close()
        table.close()
        sunks.close()
        rs.close()
