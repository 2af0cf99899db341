#!/usr/bin/env python3
"""End-to-end timing of the drop-in CLI on config-2-scale files (written to a temp dir first)."""
import os, sys, time, tempfile, gzip, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gavisunk_amd import synth, kmerpos_annot3, engine, _lib
from oracle import sunkdef

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = tempfile.mkdtemp(prefix="gs_e2e_")
t0 = time.perf_counter()
genome, rng = synth.genome_config2(20)
sunks = sunkdef.define_sunks([("ctg5M", synth.codes_to_ascii(genome))], 20)
codes = np.sort(sunks.code)
# fast writers (numpy): db and loc
A = np.frombuffer(b"ACGT", np.uint8)
def kmers_ascii(c):
    out = np.empty((len(c), 21), np.uint8); out[:, 20] = 10
    for j in range(20):
        out[:, j] = A[((c >> np.uint64(2 * (19 - j))) & np.uint64(3)).astype(np.int64)]
    return out
open(f"{d}/db", "wb").write(kmers_ascii(codes).tobytes())
ka = kmers_ascii(sunks.code)
with open(f"{d}/loc", "wb") as f:
    lines = [b"ctg5M\t%d\t%s\t%d\n" % (s, ka[i, :20].tobytes(), g) for i, (s, g) in enumerate(zip(sunks.start.tolist(), sunks.group.tolist()))]
    f.write(b"".join(lines))
rr = np.random.default_rng([20, 0])
lens = synth.lognormal_lengths(rr, n_reads, median=19000.0)
reads = synth.simulate_reads(rr, [genome], lens)
names = [f"r{i}" for i in range(n_reads)]
synth.write_fasta(f"{d}/reads.fa", names, reads)
bases = sum(len(r) for r in reads)
print(f"files in {time.perf_counter()-t0:.1f} s: {bases} bases, db {os.path.getsize(d+'/db')>>20} MiB, loc {os.path.getsize(d+'/loc')>>20} MiB, fa {os.path.getsize(d+'/reads.fa')>>20} MiB", flush=True)
subprocess.check_call(f"gzip -1 -k -c {d}/reads.fa > {d}/reads.fa.gz", shell=True)
for reads_file in ("reads.fa", "reads.fa.gz"):
    t0 = time.perf_counter(); db = engine.SunkDB(f"{d}/db", f"{d}/loc"); t_load = time.perf_counter() - t0
    t0 = time.perf_counter(); tab = engine.DeviceTable.from_db(db); t_tab = time.perf_counter() - t0
    t0 = time.perf_counter(); nb = 0
    t_parse = t_up = t_tag = t_fmt = 0.0
    carry = _lib.GS_NONE; rows = 0
    it = engine.read_batches(f"{d}/{reads_file}", max_bases=1 << 31)
    while True:
        t1 = time.perf_counter()
        try: b = next(it)
        except StopIteration: break
        t2 = time.perf_counter(); dev = engine.DeviceReads(b, db.k); t3 = time.perf_counter()
        st, hits, carry, ctr = engine.tag(tab, dev, carry); t4 = time.perf_counter()
        txt = engine.format_sunkpos(hits, b.names, db); t5 = time.perf_counter()
        t_parse += t2 - t1; t_up += t3 - t2; t_tag += t4 - t3; t_fmt += t5 - t4; rows += len(hits); nb += b.n_bases
    tot = time.perf_counter() - t0
    print(f"{reads_file}: load db+loc {t_load:.2f} s, table {t_tab:.2f} s | reads: parse+pack {t_parse:.2f} s, upload {t_up:.3f} s, tag {t_tag:.4f} s, format {t_fmt:.3f} s ({rows} rows) -> {nb/tot/1e6:.0f} Mbase/s over the read phase", flush=True)
    tab.close(); db.close()
t0 = time.perf_counter()
rc = kmerpos_annot3.main([f"{d}/reads.fa", f"{d}/db", f"{d}/loc", f"{d}/out.sunkpos"])
print(f"CLI kmerpos_annot3 reads.fa: rc {rc}, {time.perf_counter()-t0:.2f} s wall for {bases} bases")
