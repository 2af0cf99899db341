#!/usr/bin/env python3
"""Headline benchmark: ONT bases/s SUNK-tagged (k=20) on N MI355X, bit-exact hit list.

Workload at N = 1 = BASELINE.json configs[2] ("config 3"), the largest single-GPU configuration:
a CHM13-scale haplotype surrogate (24 contigs, 3.1 Gb, half of it in pasted repeat families), its EXACT SUNK set
(k = 20, ~1.5e9 keys, defined on the GPU) in an HBM-resident table, and 1 000 000 synthetic ONT reads (lognormal, N50 ~ 50 kb,
6 % errors, both strands; ~4e10 bases) PER GPU.  Everything is generated, packed and tagged in HBM (gavisunk_amd.synth_gpu);
real CHM13 / ONT data do not exist offline.  At N > 1 every rank tags its own shard of one logical input file (weak
scaling) with the table replicated.

A step = one full tagging pass over the rank's resident reads (inputs already in HBM): probe kernel, tile scans, emit
kernel, hit list to the rank's host; at N > 1 also the carry exchange (all_gather) and the RCCL gather of the hit buffers
into the root's device memory (the gather of pass i runs beside the probe kernels of pass i + 1; the last one is flushed
inside the timed region).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0: the driver contract's keys, `roofline` (probe kernel: algorithmic bytes / HIP-event time
against 8 TB/s HBM), `cpu_baseline` (the oracle's C restatement of the reference tagger on this box's host cores, on a
bounded sample), `parity` (GPU vs oracle on the first reads of the stream), `engine_bases_per_s` (host-packed words ->
hits on host, upload included, on a sample), and `config2` (the same measurements on BASELINE.json configs[1], every row
compared with the reference binary's golden output, plus `cli`: wall time of the command-line drop-in `.fa -> .sunkpos` on
the same files; `--no-config2` / `--no-cli` skip them).  `--genome-mb / --reads` shrink the workload
for rehearsals (the line then says so in `config.workload`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import gzip
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
K = 20


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.perf_counter()


def alg_bytes_per_base(h_raw: float, h_emit: float) -> float:
    """0.25 B (2-bit read stream) + 8 B (one key-slot probe per window)
    + 16 B per raw hit (payload fetch) + 12 B per emitted hit (output record)."""
    return 8.25 + 16.0 * h_raw + 12.0 * h_emit


def kernel_names(layout: str) -> str:
    return {"minimizer-hbm": "k_probe_mz"}.get(layout.split(":")[0], "k_probe")


def source_hash() -> str:
    """Hash of the kernel sources as the compiler sees them: `//` comments and blank lines do not count (so that a comment
    edit does not orphan the measured traffic figure; any code edit does)."""
    h = hashlib.sha256()
    for f in ("gs_device.hip", "gs_mz.h", "gs_mzp.h"):
        for ln in open(os.path.join(ROOT, "gavisunk_amd", "csrc", f), "r"):
            code = ln.split("//", 1)[0].rstrip()
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def traffic_for(workload_key: str):
    """HBM bytes per launch from a separate rocprofv3 --pmc run (never collected inside the bench); quoted only for the
    exact workload and kernel sources they were measured on (profiles/r04/traffic.json: one entry per GPU workload)."""
    for rnd in ("r05", "r04", "r03"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", rnd, "traffic.json")))
            e = tj[workload_key]
            if e.get("source_hash") == source_hash():
                return e["hbm_bytes_per_launch"], e.get("note")
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def timed_oracle(table, seqs, threads, target_seconds):
    """`threads` independent chunks of the sample, one oracle thread each (Snakemake-scatter style)."""
    from oracle import oracle

    buf, off = oracle.concat_reads(seqs[:32])
    for _ in range(2):  # calibrate (twice: the first pass pays the page faults of the table)
        t0 = time.perf_counter()
        table.tag(buf, off, cap=1 << 20)
        rate = max(float(off[-1]) / max(time.perf_counter() - t0, 1e-6), 1.0)  # bases/s/thread
    want = rate * target_seconds * threads
    have = float(sum(len(s) for s in seqs))
    n, acc = 0, 0
    while n < len(seqs) and acc < want:
        acc += len(seqs[n])
        n += 1
    n = max(n, min(threads, len(seqs)))
    reps = max(1, int(round(want / have))) if n == len(seqs) else 1
    packed = [oracle.concat_reads(seqs[:n][i::threads]) for i in range(threads)]

    def work(i):
        for _ in range(reps):
            table.tag(packed[i][0], packed[i][1], cap=1 << 23)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    bases = int(sum(int(p[1][-1]) for p in packed)) * reps
    return bases / dt, n, reps, bases, dt


def same_hits(gh, oh, rows_of_oracle=None):
    orow = oh["row"] if rows_of_oracle is None else rows_of_oracle[oh["row"]]
    return bool(len(gh) == len(oh) and np.array_equal(gh["read"], oh["read"]) and np.array_equal(gh["pos"], oh["pos"])
                and np.array_equal(gh["payload"], orow))


# ------------------------------------------------------------------------------------------------------------------
# the genome-scale workloads (synth_gpu.WORKLOADS): everything on the device
# ------------------------------------------------------------------------------------------------------------------
def build_workload(a, w, device, rank, world, comm, shards):
    """-> table, resident device batches [(DeviceReads, first read, read lengths)], SUNK count, group count, oracle sample.
    With an RCCL communicator only rank 0 defines the SUNKs and builds the table; the others receive the built table.
    `shards`: the read shards (synth_gpu.make_reads) this rank tags, each cut into a.batches resident device batches."""
    from gavisunk_amd import engine, synth_gpu

    t0 = time.perf_counter()
    gen = synth_gpu.make_genome(w, a.genome_mb, device)
    log(f"assembly surrogate: {gen.n_bases} bases in {len(gen.off) - 1} contigs ({time.perf_counter() - t0:.1f} s)")
    replicate = comm is not None and comm.rccl is not None
    sunks = table = None
    if rank == 0 or not replicate or getattr(a, "keep_sunks", False):
        # (the sharded second half turns payloads into contig / start / ID through the SUNK set's own arrays: every rank defines it)
        t0 = time.perf_counter()
        sunks = engine.DeviceSunkSet(gen.ptr, gen.off, K, device)
        log(f"exact SUNK set: {sunks.n} keys in {sunks.n_groups} groups ({time.perf_counter() - t0:.1f} s, gs_sunks_define_device)")
    if rank == 0 or not replicate:
        t0 = time.perf_counter()
        table = sunks.table()
        log(f"table: {table.layout} | {table.nbytes / 2**30:.1f} GiB ({time.perf_counter() - t0:.1f} s)")
    if replicate:
        t0 = time.perf_counter()
        table = comm.rccl.broadcast_table(table, root=0)
        log(f"table replicated over RCCL ({time.perf_counter() - t0:.1f} s, gs_table_broadcast)")
    batches, sample, read0 = [], None, 0
    for si, shard in enumerate(shards):
        t0 = time.perf_counter()
        rs = synth_gpu.make_reads(w, gen, a.reads, shard)
        rlen = (rs.off[1:] - rs.off[:-1]).astype(np.int64)
        n = rs.n_reads
        cuts = [n * j // a.batches for j in range(a.batches + 1)]
        for lo, hi in zip(cuts, cuts[1:]):
            if hi > lo:
                dr = engine.device_reads_from_device_ascii(rs.ptr + int(rs.off[lo]), rs.off[lo:hi + 1] - rs.off[lo], K, device)
                batches.append((dr, read0 + lo, rlen[lo:hi]))
        log(f"shard {shard}: {n} reads, {rs.n_bases} bases, N50 {synth_gpu.n50(rlen)}, packed on the device in {a.batches} batches ({time.perf_counter() - t0:.1f} s)")
        # ---- the sample the oracle walks: the first reads of the rank-0 stream, and the SUNKs that can matter for them ----
        if si == 0 and rank == 0 and not a.no_cpu_baseline and sunks is not None:
            t0 = time.perf_counter()
            from oracle import oracle

            ns = min(a.check_reads, n, cuts[1])
            seqs = rs.host_reads(0, ns)
            # every canonical window code of the sample (CPU, oracle library), then the rows of the SUNK set that hold one of
            # them (a plain set-membership kernel of the bench-support library: independent of the product's tables)
            q = np.unique(np.concatenate([oracle.canonical_windows(s if len(s) >= K else s + b"A" * (K - len(s)), K) for s in seqs]))
            L = engine.lib()
            rows = synth_gpu.select_rows(device, L.gs_sunkset_dev_code(sunks._h), sunks.n, q)
            sub = {"rows": rows, "code": synth_gpu.gather(device, L.gs_sunkset_dev_code(sunks._h), np.uint64, rows),
                   "pos": synth_gpu.gather(device, L.gs_sunkset_dev_pos(sunks._h), np.uint64, rows),
                   "group": synth_gpu.gather(device, L.gs_sunkset_dev_group(sunks._h), np.uint32, rows)}
            sample = {"seqs": seqs, "sub": sub}
            log(f"oracle sample: {ns} reads, {len(q)} distinct windows, {len(rows)} SUNKs among them ({time.perf_counter() - t0:.1f} s)")
        read0 += n
        rs.close()
    contig_names = list(gen.names)
    gen.close()
    n_sunks, n_groups = (sunks.n, sunks.n_groups) if sunks is not None else (0, 0)
    keep = None
    if sunks is not None:
        if getattr(a, "keep_sunks", False):
            keep = (sunks, contig_names)  # (the device-resident stage-2/3 leg turns payloads into contig / start / ID through its arrays)
        else:
            sunks.close()
    return table, batches, n_sunks, n_groups, sample, keep


# ------------------------------------------------------------------------------------------------------------------
# config 2 (BASELINE.json configs[1]): the pinned inputs of tests/golden/config2_full
# ------------------------------------------------------------------------------------------------------------------
def config2_leg(a, device):
    from gavisunk_amd import _lib, engine, synth
    from oracle import oracle

    L = _lib.lib()
    t0 = time.perf_counter()
    seq, sunks, names, reads = synth.config2(engine.define_sunks, seed=20, n_reads=10000, k=K)
    _, loc = np.unique(sunks.group, return_inverse=True)
    seqs = [synth.codes_to_ascii(r) for r in reads]
    n_bases = sum(len(s) for s in seqs)
    table = engine.DeviceTable(K, sunks.code, loc.astype(np.uint32), np.arange(len(sunks), dtype=np.uint32), device)
    batch = engine.ReadBatch.from_ascii(seqs)
    dev_reads = engine.DeviceReads(batch, K, device)
    log(f"config 2: {len(sunks)} SUNKs, {n_bases} bases, table {table.layout.split(':')[0]} ({time.perf_counter() - t0:.1f} s)")
    n_hits, carry, ctr = C.c_uint64(0), C.c_uint32(0), _lib.GsCounters()
    host_hits = np.zeros(1 << 20, dtype=engine.HIT_DTYPE)

    def step():
        _lib.check(L.gs_tag_launch(table._h, dev_reads._h, _lib.GS_NONE))
        _lib.check(L.gs_tag_finish(table._h, dev_reads._h, host_hits.ctypes.data, len(host_hits), C.byref(n_hits), C.byref(carry), C.byref(ctr)))

    for _ in range(3):
        step()
    L.gs_counters_reset(table._h)
    times = []
    for _ in range(20):
        t1 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t1)
    c = ctr.as_dict()
    h_raw, h_emit = c["raw_hits"] / c["windows"], c["emitted"] / c["windows"]
    probe_s = c["probe_ms"] / c["launches"] / 1e3
    bpb = alg_bytes_per_base(h_raw, h_emit)
    # engine rate: host-packed words -> hits on host, upload included
    t1 = time.perf_counter()
    d2 = engine.DeviceReads(batch, K, device)
    st, gh, _, _ = engine.tag(table, d2)
    t_engine = time.perf_counter() - t1
    d2.close()
    # every row against the reference binary's output on these inputs (tests/golden/config2_full)
    hits = host_hits[: n_hits.value]
    want = gzip.open(os.path.join(ROOT, "tests", "golden", "config2_full", "c2.sunkpos.gz"), "rb").read().split(b"\n")[:-1]
    start, group = sunks.start.tolist(), sunks.group.tolist()
    got = [b"r%d\t%d\tctg5M\t%d\t%d" % (r, p, start[w], group[w])
           for r, p, w in zip(hits["read"].tolist(), hits["pos"].tolist(), hits["payload"].tolist())]
    out = {"workload": "BASELINE.json configs[1]: single 5 Mb contig, 10 000 synthetic ONT reads, k=20 (synth.config2 seed 20)",
           "value": n_bases / float(np.median(times)), "unit": "bases/s", "ms_per_step": {"min": min(times) * 1e3, "median": float(np.median(times)) * 1e3, "max": max(times) * 1e3},
           "bases": n_bases, "sunks": int(len(sunks)), "table_layout": table.layout, "h_raw": round(h_raw, 5),
           "roofline": {"kernel": kernel_names(table.layout), "kernel_ms": round(probe_s * 1e3, 4), "alg_bytes_per_base": round(bpb, 3),
                        "achieved": round(n_bases * bpb / probe_s / 1e9, 1), "unit": "GB/s", "frac": round(n_bases * bpb / probe_s / 1e9 / HBM_PEAK_GBS, 4)},
           "engine_bases_per_s": n_bases / t_engine,
           "parity": {"against": "reference binary kmerpos_annot3 (tests/golden/config2_full/c2.sunkpos.gz)", "checked_reads": 10000,
                      "rows": len(want), "bit_exact": bool(got == want and st == 0 and same_hits(gh, hits.view(np.dtype([('read', '<u4'), ('pos', '<u4'), ('row', '<u4')]))))}}
    try:
        out["pair_stage"] = config2_chain(table, dev_reads, sunks, names, seqs, device)
    except Exception as e:  # (a measurement leg must not cost the headline)
        out["pair_stage"] = {"failed": repr(e)}
    if a.cli:
        out["cli"] = config2_cli(sunks, names, reads, n_bases, want)
    dev_reads.close()
    table.close()
    batch.close()
    return out


def config2_cli(sunks, names, reads, n_bases, want):
    """`python -m gavisunk_amd.kmerpos_annot3 reads.fq.gz db loc out` in a warm process: .fq.gz -> .sunkpos wall time."""
    import tempfile

    from gavisunk_amd import kmerpos_annot3, synth

    lut = np.frombuffer(b"ACGT", dtype=np.uint8)

    def kmers(c):
        o = np.empty((len(c), K + 1), np.uint8)
        o[:, K] = 10
        for j in range(K):
            o[:, j] = lut[((c >> np.uint64(2 * (K - 1 - j))) & np.uint64(3)).astype(np.int64)]
        return o

    with tempfile.TemporaryDirectory(prefix="gs_bench_cli_", dir="/tmp") as d:
        open(f"{d}/db", "wb").write(kmers(np.sort(sunks.code)).tobytes())
        ka = kmers(sunks.code)
        with open(f"{d}/loc", "wb") as f:
            f.write(b"".join(b"ctg5M\t%d\t%s\t%d\n" % (s, ka[i, :K].tobytes(), g)
                             for i, (s, g) in enumerate(zip(sunks.start.tolist(), sunks.group.tolist()))))
        synth.write_fasta(f"{d}/reads.fa", names, reads)
        # what the workflow's rule really feeds the tagger (tagONT.smk:17,36): `seqtk seq -F '#'` output, gzip-compressed
        t0 = time.perf_counter()
        with gzip.open(f"{d}/reads.fq.gz", "wb", compresslevel=1) as f:
            for nm, rd in zip(names, reads):
                sq = synth.codes_to_ascii(rd)
                f.write(b"@" + nm.encode() + b"\n" + sq + b"\n+\n" + b"#" * len(sq) + b"\n")
        log(f"config 2 CLI inputs written ({time.perf_counter() - t0:.1f} s for the .fq.gz)")
        res = {}
        for tag, path in (("fasta", f"{d}/reads.fa"), ("fastq_gz", f"{d}/reads.fq.gz")):
            t0 = time.perf_counter()
            rc = kmerpos_annot3.main([path, f"{d}/db", f"{d}/loc", f"{d}/out.sunkpos"])
            dt = time.perf_counter() - t0
            ok = rc == 0 and open(f"{d}/out.sunkpos", "rb").read().split(b"\n")[:-1] == want
            tm = dict(kmerpos_annot3.last_timing)
            tm.pop("rows", None)
            res[tag] = {"seconds": round(dt, 3), "cli_bases_per_s": n_bases / dt, "bit_exact": bool(ok), "seconds_by_phase": {k_: round(v, 3) for k_, v in tm.items()}}
        # the same with the prebuilt sidecar of db + loc (`--prepare`, once per sample: the rule runs the tagger once per scatter
        # chunk and haplotype on the same two files, tagONT.smk:19-37)
        if kmerpos_annot3.main(["--prepare", f"{d}/db", f"{d}/loc"]) == 0:
            for rep in range(2):
                t0 = time.perf_counter()
                rc = kmerpos_annot3.main([f"{d}/reads.fq.gz", f"{d}/db", f"{d}/loc", f"{d}/out.sunkpos"])
                dt = time.perf_counter() - t0
            ok = rc == 0 and open(f"{d}/out.sunkpos", "rb").read().split(b"\n")[:-1] == want
            tm = dict(kmerpos_annot3.last_timing)
            tm.pop("rows", None)
            res["fastq_gz_with_sidecar"] = {"seconds": round(dt, 3), "cli_bases_per_s": n_bases / dt, "bit_exact": bool(ok),
                                            "seconds_by_phase": {k_: round(v, 3) for k_, v in tm.items()}}
        res["note"] = ("`python -m gavisunk_amd.kmerpos_annot3 <reads> <db> <loc> <out>` in a warm process; fastq_gz = the format rule SUNK_annot is "
                       "given (tagONT.smk:17,36): inflated by the native parallel reader (gs_gz_*, csrc/gs_inflate.cpp), records parsed on the GPU, rows "
                       "written natively; without the sidecar the run is bound by parsing <db> and <loc> (3.7e6 text rows each) and building the table")
    return res


def cli_large_leg(a, device):
    """End to end at >= 1e8 keys (SURVEY.md section 8d measurement (iii)): the command-line drop-in on FILES -- a 200 Mb assembly
    surrogate's k-mer list and location table (1e8 rows of text each, written natively) and a read file -- as rule SUNK_annot runs
    it once per scatter chunk (tagONT.smk:19-37): parsing the two text files every time like the reference
    (kmerpos_annot3.nim:20-72), and with the prebuilt sidecar (`--prepare`: parsed rows + built table, fingerprinted)."""
    import shutil
    import tempfile

    from gavisunk_amd import engine, kmerpos_annot3, synth_gpu

    if shutil.disk_usage("/tmp").free < 48 << 30:
        return {"skipped": "less than 48 GiB free in /tmp"}
    res = {}
    with tempfile.TemporaryDirectory(prefix="gs_bench_cli_large_", dir="/tmp") as d:
        t0 = time.perf_counter()
        gen = synth_gpu.Genome(200_000_000, n_contigs=24, seed=31, device=device)
        sunks = engine.DeviceSunkSet(gen.ptr, gen.off, 20, device)
        sunks.write_files(gen.names, f"{d}/db", f"{d}/loc")
        rng = np.random.default_rng(31)
        rs = synth_gpu.ReadSet(gen, synth_gpu.lognormal_lengths(rng, 4000, median=32000.0), seed=31000)
        big = None
        if not a.no_cli_genome_scale and shutil.disk_usage("/tmp").free > 64 << 30:
            # >= 1e10 bases of reads for the genome-scale run below (measurement (iii) at scale), written as rule split_ONT writes
            # them: `seqtk seq -F '#'` FASTQ, gzip members one after the other (tagONT.smk:17; compressed by 16 threads here)
            big = synth_gpu.ReadSet(gen, synth_gpu.lognormal_lengths(rng, 250_000, median=32000.0), seed=31001)
        gen.close()
        with open(f"{d}/reads.fa", "wb") as f:
            for i, sq in enumerate(rs.host_reads(0, rs.n_reads)):
                f.write(b">r%d\n" % i + sq + b"\n")
        n_bases, n_keys = rs.n_bases, sunks.n
        rs.close()
        sunks.close()
        big_bases = 0
        if big is not None:
            from concurrent.futures import ThreadPoolExecutor

            t1 = time.perf_counter()
            big_bases, nb = big.n_bases, big.n_reads

            import struct
            import zlib

            def bgzf_member(chunk):  # bgzip's container: <= 64 KiB of text per member, its compressed size in the `BC` extra field
                c = zlib.compressobj(1, zlib.DEFLATED, -15)
                raw = c.compress(chunk) + c.flush()
                return (b"\x1f\x8b\x08\x04" + b"\0" * 6 + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1) + raw +
                        struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))

            def member(lo):
                hi = min(lo + 500, nb)
                text = b"".join(b"@g%d\n" % (lo + j) + sq + b"\n+\n" + b"#" * len(sq) + b"\n" for j, sq in enumerate(big.host_reads(lo, hi)))
                return gzip.compress(text, 1), b"".join(bgzf_member(text[at:at + 65280]) for at in range(0, len(text), 65280))

            # the same text twice: as plain multi-member gzip, and in bgzip's container -- what `rustybam fastq-split` (tagONT.smk:17)
            # writes its `.gz` outputs in, and what the GPU inflate path (csrc/gs_bgzf.hip) takes
            with open(f"{d}/big.fq.gz", "wb") as f, open(f"{d}/big.bgzf.fq.gz", "wb") as fz, ThreadPoolExecutor(16) as pool:
                for blob, zblob in pool.map(member, range(0, nb, 500)):
                    f.write(blob)
                    fz.write(zblob)
                fz.write(bgzf_member(b""))
            big.close()
            log(f"genome-scale reads: {nb} reads, {big_bases} bases -> big.fq.gz {os.path.getsize(f'{d}/big.fq.gz') / 2**30:.2f} GiB ({time.perf_counter() - t1:.1f} s)")
        log(f"CLI at {n_keys} keys: files written ({time.perf_counter() - t0:.1f} s; db {os.path.getsize(f'{d}/db') / 2**30:.1f} GiB, loc {os.path.getsize(f'{d}/loc') / 2**30:.1f} GiB)")

        phases = {}

        def run(tag):
            t1 = time.perf_counter()
            rc = kmerpos_annot3.main([f"{d}/reads.fa", f"{d}/db", f"{d}/loc", f"{d}/out.{tag}"])
            dt = time.perf_counter() - t1
            phases[tag] = {k_: round(v, 3) for k_, v in kmerpos_annot3.last_timing.items() if k_ != "rows"}
            return rc, dt

        rc1, t_plain = run("plain")
        t1 = time.perf_counter()
        rc2 = kmerpos_annot3.main(["--prepare", f"{d}/db", f"{d}/loc"])
        t_prep = time.perf_counter() - t1
        rc3, t_side = run("side")
        same = rc1 == rc2 == rc3 == 0 and open(f"{d}/out.plain", "rb").read() == open(f"{d}/out.side", "rb").read()
        scale = None
        if big_bases:
            t1 = time.perf_counter()
            rc4 = kmerpos_annot3.main([f"{d}/big.fq.gz", f"{d}/db", f"{d}/loc", f"{d}/out.big"])
            t_big = time.perf_counter() - t1
            tm = dict(kmerpos_annot3.last_timing)
            rows_big = int(tm.pop("rows", 0))
            # spot check: the first rows against a plain run over the head of the file's text (first member = first 500 reads)
            head = gzip.GzipFile(fileobj=open(f"{d}/big.fq.gz", "rb")).read(20_000_000)
            cut = head.rfind(b"\n@g")
            open(f"{d}/head.fq", "wb").write(head[:cut + 1])
            rc5 = kmerpos_annot3.main([f"{d}/head.fq", f"{d}/db", f"{d}/loc", f"{d}/out.head"])
            hd = open(f"{d}/out.head", "rb").read()
            last_name = hd[hd.rfind(b"\n", 0, len(hd) - 1) + 1:].split(b"\t")[0] if hd else b""
            with open(f"{d}/out.big", "rb") as fb:
                hb = fb.read(len(hd) + 4096)
            # (the carry of the head run and of the full run agree up to the head's last read)
            ok_head = rc4 == rc5 == 0 and len(hd) > 1000 and hb[:hd.rfind(b"\n" + last_name + b"\t") + 1] == hd[:hd.rfind(b"\n" + last_name + b"\t") + 1]
            scale = {"read_bases": int(big_bases), "fq_gz_GiB": round(os.path.getsize(f"{d}/big.fq.gz") / 2**30, 2), "rows_out": rows_big,
                     "sunkpos_MiB": round(os.path.getsize(f"{d}/out.big") / 2**20, 1), "seconds": round(t_big, 2), "cli_bases_per_s": big_bases / t_big,
                     "seconds_by_phase": {k_: round(v, 2) for k_, v in tm.items()},
                     "head_of_output_equals_a_run_over_the_head_of_the_file": bool(ok_head),
                     "note": "`python -m gavisunk_amd.kmerpos_annot3 big.fq.gz db loc out` with the sidecar: multi-member .fq.gz (quals '#') inflated by the native parallel reader "
                             "(gs_gz_*), text blocks copied to the device and parsed there, rows written natively (gs_rowwriter_*).  read_or_inflate_text and "
                             "copy_text_to_device run on the reader thread beside the others; wait_for_reader is what the main thread waits for them"}

            # the same reads in bgzip's container: read, inflated (on the GPU), parsed, tagged, written -- the bytes of the run above
            bgzf_runs = []
            for rep in range(2):
                t1 = time.perf_counter()
                rc6 = kmerpos_annot3.main([f"{d}/big.bgzf.fq.gz", f"{d}/db", f"{d}/loc", f"{d}/out.bgzf"])
                bgzf_runs.append(time.perf_counter() - t1)
            tmz = dict(kmerpos_annot3.last_timing)
            tmz.pop("rows", None)

            def same_file(x, y):
                if os.path.getsize(x) != os.path.getsize(y):
                    return False
                with open(x, "rb") as fx, open(y, "rb") as fy:
                    while True:
                        bx, by = fx.read(1 << 24), fy.read(1 << 24)
                        if bx != by:
                            return False
                        if not bx:
                            return True

            t_z = min(bgzf_runs)
            scale["bgzf"] = {"fq_gz_GiB": round(os.path.getsize(f"{d}/big.bgzf.fq.gz") / 2**30, 2), "seconds": round(t_z, 2), "seconds_of_each_run": [round(x, 2) for x in bgzf_runs],
                             "cli_bases_per_s": big_bases / t_z, "seconds_by_phase": {k_: round(v, 2) for k_, v in tmz.items()},
                             "same_bytes_as_the_gzip_run": bool(rc6 == 0 and "read_and_inflate_on_device" in tmz and same_file(f"{d}/out.big", f"{d}/out.bgzf")),
                             "note": "the same reads in bgzip's container (members of <= 64 KiB of text with their compressed size in the header: what `rustybam fastq-split` "
                                     "writes, tagONT.smk:17): the compressed bytes go to the device, k_bgzf_inflate decodes a member per lane, CRC-32 and ISIZE checked there, "
                                     "the text block is parsed where it lies (csrc/gs_bgzf.hip); seconds = the faster of two runs"}

        res = {"keys": int(n_keys), "read_bases": int(n_bases), "rows_out": open(f"{d}/out.plain", "rb").read().count(b"\n"),
               "parse_and_build_every_run": {"seconds": round(t_plain, 2), "cli_bases_per_s": n_bases / t_plain, "seconds_by_phase": phases.get("plain")},
               "prepare_once": {"seconds": round(t_prep, 2), "sidecar_GiB": round(os.path.getsize(f"{d}/loc.gsidx") / 2**30, 2)},
               "with_sidecar": {"seconds": round(t_side, 2), "cli_bases_per_s": n_bases / t_side, "seconds_by_phase": phases.get("side")},
               "same_bytes": bool(same),
               **({"cli_genome_scale": scale} if scale else {}),
               "note": "`python -m gavisunk_amd.kmerpos_annot3 reads.fa db loc out` in a warm process on a 200 Mb surrogate's files (the reference parses db and loc on every "
                       "run, once per scatter chunk: kmerpos_annot3.nim:20-72, tagONT.smk:19-37); with_sidecar = after `--prepare db loc` (parsed location rows + built table in "
                       "loc.gsidx, used only while its fingerprints of db and loc hold)"}
    return res


def engine_leg(a, table, batches, device, target_bases=float("inf")):
    """SURVEY.md section 8(d) measurement (ii): host-packed reads in PINNED host memory -> hits on the host, through the
    streaming engine (gs_stream_*): upload of batch i + 1 overlapped with the pass of batch i, hit lists fetched on the copy
    stream.  The packed words are the resident batches' own (copied to the host once, outside the timed region)."""
    from gavisunk_amd import _lib, engine

    L = _lib.lib()
    host = []
    total = 0
    for dr, _, rlen in batches:
        if total >= target_bases:
            break
        word_off = np.zeros(len(rlen) + 1, dtype=np.uint64)
        word_off[1:] = np.cumsum((rlen + 31) // 32 + 1)
        pw = engine.PinnedWords(int(word_off[-1]))
        _lib.check(L.gs_reads_copy_words(dr._h, 0, int(word_off[-1]), pw.words.ctypes.data))
        host.append((pw, word_off, rlen.astype(np.uint32)))
        total += int(rlen.sum())
    max_words = max(len(h[0].words) for h in host)
    max_reads = max(len(h[2]) for h in host)
    t0 = time.perf_counter()
    st = engine.Stream(table, max_words, max_reads, depth=3)
    t_create = time.perf_counter() - t0
    rows = 0
    for rep in range(2):  # the first round warms the slots (first touch of the arenas), the second is timed
        rows = 0
        flying = 0
        _lib.check(L.gs_device_synchronize(device))
        t0 = time.perf_counter()
        for pw, word_off, ln in host:
            if flying == 2:
                rows += len(st.next(copy=False)[1])
                flying -= 1
            st.submit(pw.words, word_off, ln)
            flying += 1
        while flying:
            rows += len(st.next(copy=False)[1])
            flying -= 1
        dt = time.perf_counter() - t0
    st.close()
    for pw, _, _ in host:
        pw.close()
    return {"engine_bases_per_s": total / dt,
            # "(ii) engine (host-packed reads in pinned memory -> hits on host); the headline is (ii)".  `value`
            # stays the resident-reads step (what the driver compares from round to round); this key is (ii) by name
            "headline_per_survey_8d": {"value": total / dt, "unit": "bases/s", "pcie_GBps_of_packed_words": total / 4 / dt / 1e9,
                                       "what": "measurement (ii): host-packed 2-bit reads in pinned host memory -> rows in host memory, every batch of the pass "
                                               "(uploads, passes and row downloads overlapped by gs_stream_*); bound by the host link, not by the kernel"},
            "engine_note": f"{len(host)} batches, {total} bases of host-packed reads in pinned memory (gs_host_alloc) -> rows on the host through "
                           f"gs_stream_submit / gs_stream_next (3 slots; upload, pass and hit-list download overlapped), {dt * 1e3:.1f} ms, {rows} rows; "
                           f"one-off slot set-up {t_create * 1e3:.0f} ms not included"}


def self_launch(n: int, script: str = None, args=None) -> int:
    """`python bench.py --gpus N` without a launcher: the N ranks are started as CHILD processes through
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1), rank 0's JSON line is relayed to this process's stdout
    and the children's exit code is passed on.  Called before anything in this process has loaded the HIP runtime; the
    children are never exec'ed over it.  (The reference's counterpart is Snakemake's scatter: tagONT.smk:2-37,112-131.)"""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(sys.argv[1:] if args is None else args)
    print(f"[bench] self-launch: {' '.join(cmd)}", file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=os.getcwd())
    line = None
    for raw in p.stdout:  # (the ranks keep stdout clean: rank 0 writes exactly one line there)
        txt = raw.decode("utf-8", "replace").rstrip("\n")
        if txt.startswith("{") and txt.endswith("}"):
            line = txt
        elif txt:
            print(txt, file=sys.stderr, flush=True)
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if rc == 0 and line is None:
        print("[bench] self-launch: the ranks exited 0 without a JSON line", file=sys.stderr)
        return 1
    return rc


def chain_once(table, batches, locrows, contig_names, device, rs=None):
    """The device-resident second half over one pass's rows: (rows gathered device to device unless `rs` holds them already) ->
    gs_rowset_diag_filter -> gs_rowset_keep_filtered -> gs_rowset_bad_keys -> gs_rowset_pair_stage.  Both haplotype roles of the
    bad-SUNK step are played by this one read set (every contig counts as its own haplotype's).  -> (seconds by phase, results)."""
    from gavisunk_amd import _lib, engine

    t = {}
    n_reads = sum(len(b[2]) for b in batches)
    rlen = np.concatenate([b[2] for b in batches])
    if rs is None:
        t0 = time.perf_counter()
        rs = engine.RowSet(device, sum(engine.hits_on_device(table, dr) for dr, _, _ in batches))
        for dr, r0, _ in batches:
            rs.append_pass(table, dr, r0, locrows)
        t["rows_from_hits"] = time.perf_counter() - t0
    rows_in = rs.n
    t0 = time.perf_counter()
    _, bc, bg, _, kept = rs.diag_filter(contig_names, np.ones(len(contig_names), np.uint8))
    t["diag_filter"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    rs.keep_filtered()
    t["step2_compaction"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    key_hap = np.full(max(locrows.n_locs, 1), 3, np.uint8)
    bad, meancov, d_bad = engine.RowSet.bad_keys(rs, rs, key_hap[: locrows.n_locs], keep_on_device=True)
    t["bad_sunks"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    # (synthetic names are numbered so that their sorted order is the stream order: rank = read index)
    res = rs.pair_stage(np.arange(n_reads, dtype=np.uint32), (rlen >= 10000).astype(np.uint8), d_key_bad=d_bad)
    t["pair_stage"] = time.perf_counter() - t0
    engine.free_device(d_bad, device)
    rs.close()
    c = res["counters"].tolist()
    summary = {"rows_in": int(rows_in), "reads": int(n_reads), "reads_with_a_best_contig": int((bc != _lib.GS_NONE).sum()),
               "rows_after_diag_filter": int(kept), "bad_sunk_groups": int(bad.sum()), "mean_coverage": meancov[0],
               "pair_stage_counters": dict(zip(("rows_in", "not_bad", "in_reads_with_2_ids", "deduplicated", "in_long_reads", "read_groups", "tsv_rows"), c[:7])),
               "tsv_rows": int(len(res["row_id"])), "validated_intervals": int(len(res["region_start"]))}
    return t, summary


def chain_leg(table, batches, locrows, contig_names, device):
    """`pair_stage` key: the second half alone over the hit list of the LAST pass of every resident batch (still in device memory),
    twice: the first call also pays for the device pool's first allocations (every later pass of a run reuses them), the second is
    the steady state a run of many passes sees and the one reported; `first_call_seconds_total` keeps the other."""
    t_cold, _ = chain_once(table, batches, locrows, contig_names, device)
    t, summary = chain_once(table, batches, locrows, contig_names, device)
    total = sum(t.values())
    rows_in = summary["rows_in"]
    moved = rows_in * 6 * 4 * 2  # the six uint32 row columns written once (rows from hits) and read once (filters)
    out = {"seconds": {k_: round(v, 4) for k_, v in t.items()}, "seconds_total": round(total, 4), "first_call_seconds_total": round(sum(t_cold.values()), 4),
           "rows_per_s": rows_in / total,
           "roofline": {"bound": "hbm", "achieved": moved / total / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": moved / total / 8e12,
                        "basis": "the six uint32 row columns written once and read once (48 B per row) / seconds_total: the chain is a string of per-read "
                                 "kernels over 1e6 reads and of small sorts, latency- and launch-bound, nowhere near a bandwidth roofline"},
           "path": "hit list in HBM -> gs_rowset_append_pass -> gs_rowset_diag_filter (csrc/gs_diag.hip: a wavefront per read, LDS) -> gs_rowset_keep_filtered -> "
                   "gs_rowset_bad_keys -> gs_rowset_pair_stage (csrc/gs_chain.hip): nothing but the final TSV / BED rows and per-read verdicts crosses PCIe"}
    out.update(summary)
    return out


def both_halves_leg(table, batches, locrows, contig_names, device, steps, warmup, n_bases):
    """`engine_both_halves` key: the engine north_star names -- tagging AND the inter-SUNK-distance half -- as ONE timed loop.  A step
    = one tagging pass over every resident batch through the streaming engine, each batch's rows gathered into a row set as soon
    as its pass has settled (chain stream, beside the next batch's probe kernel), and the second half of the PREVIOUS pass's row
    set run once this pass is queued: the chain (one non-blocking stream, latency-bound per-read kernels and
    small sorts) runs beside the probe kernels (request-bound) of the tagger's stream.  The last pass's chain is flushed inside
    the timed region.  -> bases/s over both halves."""
    from gavisunk_amd import engine

    depth = min(len(batches), 16)
    stream = engine.Stream(table, 0, 0, depth=depth)
    state = {"prev": None, "summary": None, "chain_s": []}

    def run_chain(rs):
        t0 = time.perf_counter()
        _, summary = chain_once(table, batches, locrows, contig_names, device, rs=rs)
        state["chain_s"].append(time.perf_counter() - t0)
        state["summary"] = summary

    inflight = []  # (row set the batch's rows go to, batch, first read), oldest first: the engine is not drained between passes

    def take_oldest():
        stream.next(copy=False)
        rs_of, d0, b0 = inflight.pop(0)
        rs_of.append_pass(table, d0, b0, locrows)

    def one_pass():
        rs = engine.RowSet(device, state["summary"]["rows_in"] if state["summary"] else 1 << 20)
        for dr, r0, _ in batches:
            # a batch goes into its next pass once the rows of its previous one are in their row set (the pass overwrites the hit list)
            while any(x[1] is dr for x in inflight) or len(inflight) == depth:
                take_oldest()
            stream.submit_reads(dr)
            inflight.append((rs, dr, r0))
        # this pass is queued on the tagger's stream (its last batches behind the previous pass's): the previous pass's row set is
        # complete -- its second half now, on its own stream; the host waits in it, the tagger does not (with three batches queued
        # it ran dry: a chain beside probe kernels takes 3-4 x as long as alone)
        if state["prev"] is not None:
            while any(x[0] is state["prev"] for x in inflight):
                take_oldest()
            run_chain(state["prev"])
        state["prev"] = rs

    def flush():
        while inflight:
            take_oldest()
        if state["prev"] is not None:
            run_chain(state["prev"])
            state["prev"] = None

    L = _lib_handle()
    for _ in range(warmup):
        one_pass()
    flush()
    L.gs_device_synchronize(device)
    state["chain_s"] = []
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    flush()
    L.gs_device_synchronize(device)
    dt = time.perf_counter() - t0
    stream.close()
    return {"value": n_bases * steps / dt, "unit": "bases/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "chain_ms_host_blocked": [round(x * 1e3, 1) for x in state["chain_s"]],
            "summary_of_last_pass": state["summary"],
            "what": "tagging pass (streaming engine, 8 resident batches) + rows from hits + diag_filter_v3 + step 2 + bad SUNKs + process-by-contig "
                    "(pair test, component, contig-wide union) per step; the second half of pass i runs on its own stream beside the probe kernels of pass i + 1; "
                    "the last pass's second half is inside the timed region"}


def rows_text_leg(table, batches, locrows, sunks, contig_names, device, sample_rows):
    """`cli_config3_rows` key: ONE pass's hit list (every resident batch) written as the tagger's output file -- formatted on the
    device (csrc/gs_format.hip), 32 MiB pieces through pinned buffers, pwrite on host threads.  Names are the synthetic reads'
    (`r0000042`).  The head of the file against the oracle's rows of the parity sample, the tail against the last hits of the last
    batch looked up row by row."""
    from gavisunk_amd import engine
    import tempfile

    d = tempfile.mkdtemp(prefix="gsrows_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(d, "pass.sunkpos")
    names = []
    for dr, r0, rl in batches:
        blob = "".join(f"r{r0 + i:07d}" for i in range(len(rl))).encode()
        ln = np.array([len(f"r{r0 + i:07d}") for i in (0, len(rl) - 1)])
        assert ln[0] == ln[1]  # (fixed width within a batch)
        w = int(ln[0])
        names.append((np.frombuffer(blob, np.uint8), np.arange(len(rl), dtype=np.uint64) * w, np.full(len(rl), w, np.uint32)))
    try:
        t0 = time.perf_counter()
        with engine.RowWriter(path) as out:
            for (dr, r0, rl), nm in zip(batches, names):
                out.write_device(table, dr, locrows, contig_names, names=nm)
            n_rows = out.close()
        dt = time.perf_counter() - t0
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            head = f.read(1 << 22)
            f.seek(max(size - (1 << 16), 0))
            tail = f.read()
        # head: the oracle's rows of the parity sample (read, pos, loc row) as text
        ok_head = None
        if sample_rows is not None and len(sample_rows["read"]):
            k_ = min(len(sample_rows["read"]), 2000)
            exp = []
            for rd, ps, p in zip(sample_rows["read"][:k_].tolist(), sample_rows["pos"][:k_].tolist(), sample_rows["row"][:k_].astype(np.int64).tolist()):
                pp = int(sunks.pos(p, p + 1)[0])
                exp.append(f"r{rd:07d}\t{ps}\t{contig_names[pp >> 32]}\t{pp & 0xFFFFFFFF}\t{int(sunks.group(p, p + 1)[0])}\n")
            exp = "".join(exp).encode()
            ok_head = head[: len(exp)] == exp
        # tail: the last hits of the last batch (its host mirror), looked up row by row
        dr, r0, rl = batches[-1]
        hv = engine.hits_view(table, dr)
        last = hv[-200:]
        exp = []
        for rd, ps, p in zip(last["read"].tolist(), last["pos"].tolist(), last["payload"].tolist()):
            pp = int(sunks.pos(p, p + 1)[0])
            exp.append(f"r{r0 + rd:07d}\t{ps}\t{contig_names[pp >> 32]}\t{pp & 0xFFFFFFFF}\t{int(sunks.group(p, p + 1)[0])}\n")
        exp = "".join(exp).encode()
        ok_tail = tail.endswith(exp) and len(exp) > 0
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    return {"rows": int(n_rows), "bytes": int(size), "seconds": round(dt, 3), "rows_per_s": n_rows / dt, "GB_per_s": size / dt / 1e9,
            "head_equals_oracle_rows": ok_head, "tail_equals_hit_list": bool(ok_tail), "file": "tmpfs (/dev/shm)" if d.startswith("/dev/shm") else "tmp dir",
            "path": "hit list in HBM -> k_row_len -> exclusive scan -> k_row_write (a wave stages 64 lines in LDS) -> 32 MiB pinned pieces -> pwrite (csrc/gs_format.hip)"}


def _lib_handle():
    from gavisunk_amd import _lib
    return _lib.lib()


def sharded_chain_leg(comm, table, batches, locrows, contig_names, device, read_lo):
    """The second half at N > 1 (`--sharded-chain`): every rank runs the device chain over ITS hit lists (they never leave their
    GPU); the Table capacity by all-gather, the reads-per-SUNK counters by all-reduce, the TSV rows merged on rank 0
    (gavisunk_amd.dist.sharded_second_half).  Timed between barriers, max over ranks.  -> dict on rank 0, None elsewhere."""
    from gavisunk_amd import _lib, engine
    from gavisunk_amd import dist as gdist

    L = _lib.lib()
    n_local = sum(len(b[2]) for b in batches)
    rlen = np.concatenate([b[2] for b in batches])
    comm.barrier()
    _lib.check(L.gs_device_synchronize(device))
    t0 = time.perf_counter()
    rs = engine.RowSet(device, 1 << 20)
    for dr, r0, _ in batches:
        rs.append_pass(table, dr, r0, locrows)
    rows_in = rs.n
    key_hap = np.full(max(locrows.n_locs, 1), 3, np.uint8)[: locrows.n_locs]
    # (synthetic names are numbered so that their sorted order is the stream order: global rank = global read index)
    local, merged = gdist.sharded_second_half(comm, rs, contig_names, np.ones(len(contig_names), np.uint8), key_hap,
                                              (read_lo + np.arange(n_local)).astype(np.uint32), (rlen >= 10000).astype(np.uint8), read_lo)
    _lib.check(L.gs_device_synchronize(device))
    dt = comm.all_max(time.perf_counter() - t0)
    rs.close()
    rows_all = int(comm.all_sum(int(rows_in)))
    kept_all = int(comm.all_sum(int(local["rows_kept"])))
    if merged is None:
        return None
    return {"rows_in": rows_all, "ranks": comm.world, "seconds_total": round(dt, 4), "rows_per_s": rows_all / dt, "rows_after_diag_filter": kept_all,
            "bad_sunk_groups": int(local["bad"].sum()), "mean_coverage": local["meancov"][0], "tsv_rows": int(len(merged["row_id"])),
            "validated_intervals": int(len(merged["region_start"])),
            "path": "per rank: hit list in HBM -> rows -> diag filter -> step 2 -> bad SUNKs -> pair stage (csrc/gs_chain.hip); across ranks: Table "
                    "capacity (all-gather, 8 B), reads per SUNK (all-reduce, n_keys x 4 B), TSV rows to the root (20 B per row) + union there"}


def config2_chain(table, dev_reads, sunks, names, seqs, device):
    """config 2 end to end through the device chain (the `pair_stage` key): tagger rows in HBM -> diag filter -> step 2 -> bad
    SUNKs -> pair stage -> union.  Checked against the reference binary's own diag output on these inputs
    (tests/golden/config2_full/c2.diag.gz) and against the oracle's restatement of the two Python scripts."""
    from gavisunk_amd import _lib, engine
    from oracle import oracle

    class _Rows:
        pass

    L = _lib.lib()
    _, loc = np.unique(sunks.group, return_inverse=True)
    db = _Rows()
    db.row_contig, db.row_start, db.row_group, db.row_loc = (np.zeros(len(sunks), np.uint32), sunks.start.astype(np.int64),
                                                            sunks.group.astype(np.int64), loc.astype(np.uint32))
    lr = engine.LocRows.from_db(db, device)
    n_hits = C.c_uint64(0)
    _lib.check(L.gs_tag_launch(table._h, dev_reads._h, _lib.GS_NONE))
    _lib.check(L.gs_tag_finish(table._h, dev_reads._h, None, 0, C.byref(n_hits), None, None))
    rlen = np.array([len(q) for q in seqs], dtype=np.int64)
    order = np.argsort(np.array(names, dtype=object), kind="stable")
    rank = np.empty(len(names), np.uint32)
    rank[order] = np.arange(len(names), dtype=np.uint32)
    best = None
    for _ in range(3):  # (first round: first touch of the allocations)
        t0 = time.perf_counter()
        rs = engine.RowSet(device, n_hits.value)
        rs.append_pass(table, dev_reads, 0, lr)
        rows_in = rs.n
        seg_name, bc, bg, bd, kept = rs.diag_filter(["ctg5M"], [1])
        rs.keep_filtered()
        bad, meancov, d_bad = engine.RowSet.bad_keys(rs, rs, np.full(lr.n_locs, 3, np.uint8), keep_on_device=True)
        res = rs.pair_stage(rank, (rlen >= 10000).astype(np.uint8), d_key_bad=d_bad)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        step2 = rs.download()
        engine.free_device(d_bad, device)
        rs.close()
    lr.close()
    # ---- parity
    diag = "".join(f"{names[n]}\tctg5M\t{g}\t{chr(d_)}\t{g}\n" for n, c_, g, d_ in zip(seg_name.tolist(), bc.tolist(), bg.tolist(), bd.tolist()) if c_ != _lib.GS_NONE)
    want_diag = gzip.open(os.path.join(ROOT, "tests", "golden", "config2_full", "c2.diag.gz"), "rb").read().decode()
    text2 = "".join(f"{names[n]}\t{p}\tctg5M\t{s_}\t{g}\n" for n, p, s_, g in zip(*(step2[k_].tolist() for k_ in ("name", "pos", "start", "id"))))
    fai = "ctg5M\t5000000\n"
    t1 = time.perf_counter()
    ebad = oracle.badsunks(fai, fai, text2, text2)
    gid = {}
    for k_, g in zip(loc.tolist(), sunks.group.tolist()):
        gid[k_] = g
    got_bad = {f"ctg5M:{gid[int(k_)]}" for k_ in np.nonzero(bad)[0]}
    rlen_text = "".join(f"{n}\t{l}\n" for n, l in zip(names, rlen.tolist()))
    etsv, ebed = oracle.process_by_contig(text2, rlen_text, "".join(b_ + "\n" for b_ in sorted(ebad)))
    t_oracle = time.perf_counter() - t1
    tsv = "".join(f"{i}\t{names[n]}\n" for i, n in zip(res["row_id"].tolist(), res["row_name"].tolist()))
    bed = "".join(f"ctg5M\t{s_}\t{e}\n" for s_, e in zip(res["region_start"].tolist(), res["region_end"].tolist()))
    return {"workload": "config 2: the tagger's rows (still in HBM) -> diag_filter_v3 -> diag_filter_step2 -> badsunks -> process-by-contig, on the device (csrc/gs_chain.hip)",
            "rows_in": int(rows_in), "reads": len(names), "seconds": round(best, 4), "rows_per_s": rows_in / best,
            "rows_after_diag_filter": int(kept), "bad_sunk_groups": int(bad.sum()), "tsv_rows": int(len(res["row_id"])),
            "validated_intervals": int(len(res["region_start"])),
            "reference": {"process_by_contig_seconds": 34.0, "diag_filters_seconds": 0.16, "reads": 6504,
                          "where": "BASELINE.md section 2: the reference's own script (pandas 2.x, graph_tool stand-in) and Nim binaries on config-2-scale inputs in the build container, one core"},
            "oracle_seconds_same_inputs": round(t_oracle, 2),
            "parity": {"diag_equals_reference_binary": bool(diag == want_diag), "bad_sunks_equal_oracle": bool(got_bad == ebad),
                       "tsv_equals_oracle": bool(tsv == etsv), "bed_equals_oracle": bool(bed == ebed)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "config3", "diploid", "config5"],
                    help="auto = config3 (BASELINE.json configs[2]) on one GPU, diploid (configs[3]: fixed 8 shards = ~1.2e11 bases over the ranks) on several")
    ap.add_argument("--reads", type=int, default=None, help="reads per shard (default: the workload's own)")
    ap.add_argument("--genome-mb", type=float, default=None, help="assembly size in Mb per haplotype (default: the workload's own)")
    ap.add_argument("--shards", type=int, default=None, help="diploid: read shards per rank (default 8 / ranks at N > 1, 1 at N = 1)")
    ap.add_argument("--batches", type=int, default=None, help="resident device batches per shard (default 8)")
    ap.add_argument("--group-len", type=int, default=0, help="haploid workloads: an interspersed 44-base element every N bases of unique sequence, so that "
                    "location groups are ~N SUNKs long (the default surrogate: ~12 000); a surrogate-realism sweep, not the headline")
    ap.add_argument("--k", type=int, default=20, help="SUNK length (headline: 20; 21 / 25 = BASELINE.json configs[4])")
    ap.add_argument("--check-reads", type=int, default=1000, help="reads of the stream the oracle walks for the parity leg")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-time budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (parity + CPU baseline)")
    ap.add_argument("--no-config2", action="store_true", help="skip the config-2 extra key")
    ap.add_argument("--no-engine", action="store_true", help="skip the streaming-engine leg")
    ap.add_argument("--no-chain", action="store_true", help="skip the device-resident stage-2/3 leg (`pair_stage` key)")
    ap.add_argument("--no-sharded-chain", action="store_true", help="N > 1: skip the read-sharded second half (`pair_stage` key)")
    ap.add_argument("--sharded-chain-timeout", type=float, default=240.0,
                    help="N > 1: seconds after which the read-sharded second half is given up: the line goes out without it and every rank exits 0")
    ap.add_argument("--sharded-chain", action="store_true",
                    help="N > 1: also run the read-sharded second half (dist.sharded_second_half: every rank filters and pair-tests its own hit list, "
                         "counters by all-reduce, TSV rows merged on the root) -> `pair_stage` key; every rank then defines the SUNK set itself")
    ap.add_argument("--rehearse-extra", action="store_true", help="run the extra workload key even on a reduced (--genome-mb / --reads) rehearsal")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra workload key (N = 1: one diploid shard; N > 1: config-3 weak scaling)")
    ap.add_argument("--cli", action="store_true", help="(default now) config-2 leg: also time the command-line drop-in")
    ap.add_argument("--no-cli", action="store_true", help="config-2 leg without the command-line timing")
    ap.add_argument("--no-cli-genome-scale", action="store_true", help="skip the >= 1e10-base .fq.gz run of the command-line timing")
    ap.add_argument("--no-cli-large", action="store_true", help="skip the command-line timing at 1e8 keys (files of a 200 Mb surrogate, with / without the sidecar)")
    a = ap.parse_args()
    a.cli = not a.no_cli

    # stdout carries exactly ONE line, the JSON: everything else any library prints there (gloo announces its connections on
    # stdout from C++) goes to stderr -- fd 1 is pointed at fd 2 until the line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    global K
    K = a.k
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves (nothing has touched the GPU yet in this process)
            os.dup2(real_stdout, 1)
            sys.exit(self_launch(a.gpus))
        a.gpus = world
    # tuning / layout knobs change what is measured: refuse them (GAVISUNK_DEVICE and the library override are harmless)
    knobs = sorted(v for v in os.environ if v.startswith("GAVISUNK_") and v not in ("GAVISUNK_DEVICE", "GAVISUNK_BENCH_BACKEND"))
    if knobs and not os.environ.get("GAVISUNK_BENCH_ALLOW_KNOBS"):
        sys.exit(f"bench.py: unset {knobs} (tuning knobs change the measured path); GAVISUNK_BENCH_ALLOW_KNOBS=1 for experiments")

    from gavisunk_amd import _lib, engine, synth_gpu

    L = _lib.lib()
    if L.gs_device_count() < 1:
        sys.exit("bench.py: no HIP device (gavisunk_amd has no CPU fallback)")
    device = local_rank % L.gs_device_count()

    comm = None
    data_plane = "single GPU"
    if world > 1:
        from gavisunk_amd import dist as gdist

        # control plane: torch.distributed over gloo (rendezvous, barrier, scalars); data plane: RCCL behind the C-ABI
        # (gs_comm_*: table broadcast, phase-1 summaries, hit gather).  torch never touches the GPU in this process: its wheel
        # bundles a second HIP runtime.
        want_rccl = os.environ.get("GAVISUNK_BENCH_BACKEND", "nccl") == "nccl"
        comm = gdist.Comm(device, rccl=False)
        data_plane = "gloo, host-staged (GAVISUNK_BENCH_BACKEND=gloo: rehearsal)"
        if want_rccl:
            from gavisunk_amd import rccl as grccl

            err = ""
            try:
                comm.rccl = grccl.RcclComm(comm, device)
            except Exception as e:  # the run still measures the path; the line says which transport carried it
                err = str(e)
            ok = comm.all_sum(1 if comm.rccl is not None else 0) == world
            if not ok:
                if comm.rccl is not None:
                    comm.rccl.close()
                comm.rccl = None
                data_plane = f"gloo, host-staged (RCCL did not start on every rank: {err or 'another rank failed'})"
            else:
                data_plane = "RCCL over xGMI (gs_table_broadcast, gs_comm_allgather_u64, gs_gather_hits)"

    def run_workload(wname, shards, batches_per_shard, steps, warmup, want_sample, want_chain=False):
        """Builds the workload, times `steps` passes, -> dict of raw results (rank 0: everything; others: timings only)."""
        w = dict(synth_gpu.WORKLOADS[wname])
        if a.group_len and w["kind"] == "haploid":
            w["group_len"] = a.group_len
            w["title"] += f" with an interspersed element every {a.group_len} bases (location groups of ~{a.group_len} SUNKs)"
        a.batches = batches_per_shard
        a.keep_sunks = bool(want_chain)
        table, batches, n_sunks, n_groups, sample, kept_sunks = build_workload(a, w, device, rank, world, comm, shards)
        if not want_sample:
            sample = None
        n_bases = int(sum(int(b[2].sum()) for b in batches))
        rlen_all = np.concatenate([b[2] for b in batches]) if batches else np.zeros(0, np.int64)
        nb = len(batches)
        last = {"rows": 0, "raw": 0, "emitted": 0, "windows": 0, "tiles": 0, "fallback": 0}
        gathered = {"rows": None}
        stream = engine.Stream(table, 0, 0, depth=3)  # (N > 1: for the no-exchange leg)

        # batches in flight on the streaming engine, oldest first: (pass id, index of the batch in its pass).  The engine is a
        # PIPELINE: a pass's last batches are still being probed / emitted / copied to the host when the next pass's first batches are
        # submitted, exactly as consecutive blocks of one long read file are (the command line never drains between blocks either);
        # `drain_single` empties it -- after the K timed steps, inside the timed region.
        flying = []
        pass_tot = {}
        pass_id = [0]
        head_box = [None]

        def take_oldest(keep_head_of=None):
            st_, hits, _, c = stream.next(copy=False)
            pid, bi = flying.pop(0)
            if keep_head_of == pid and bi == 0:
                head_box[0] = hits.copy()
            tot = pass_tot[pid]
            for k_, v in (("rows", len(hits)), ("raw", c["raw_hits"]), ("emitted", c["emitted"]), ("windows", c["windows"]), ("tiles", c["tiles"]), ("fallback", c["fallback_tiles"])):
                tot[k_] += v
            tot["done"] += 1
            if tot["done"] == len(batches):  # the pass is complete: its totals are the `last` ones
                del tot["done"]
                last.update(tot)
                del pass_tot[pid]

        def drain_single(keep_head_of=None):
            while flying:
                take_oldest(keep_head_of)

        def step_single(keep_head=False):
            """The rank's resident batches through the streaming engine, no exchange with other ranks (the N = 1 step; at N > 1 the
            reference point `no_exchange`: what the same ranks would do if they never talked).  Returns with up to three batches
            still in flight (drain_single); keep_head: drains, and returns the rows of the pass's first batch."""
            pid = pass_id[0]
            pass_id[0] += 1
            pass_tot[pid] = dict.fromkeys(last, 0)
            pass_tot[pid]["done"] = 0
            for bi, (dr, _, _) in enumerate(batches):
                if len(flying) == 3:
                    take_oldest(pid if keep_head else None)
                stream.submit_reads(dr)
                flying.append((pid, bi))
            if keep_head:
                drain_single(pid)
                return head_box[0]
            return None

        def step(keep_head=False, gather_to_host=False):
            """One pass over all the rank's resident reads.  N = 1: the batches pipelined through gs_stream_submit_reads (the
            carry goes from pass to pass on the device; every batch's rows are fetched to pinned host memory on the copy stream
            while the next batch is probed).  N > 1: phase 1 of every batch, ONE exchange of the rank summaries, phase 2 with the
            resolved carries, then the RCCL gather of every batch's rows into the root's device memory."""
            if world == 1:
                return step_single(keep_head)
            tot = dict.fromkeys(last, 0)
            # the probe kernels of every batch are queued first; while they run, the rows of the PREVIOUS pass (still in device
            # memory: phase 2 of this pass has not started) go to the root over xGMI -- the gather of pass i overlaps the probing
            # of pass i + 1, and the last pass's gather is flushed inside the timed region (flush_gathers below)
            # (a rank that fails here still walks the same collectives as the others -- the gather of the previous pass, the exchange --
            # and says so IN the exchange: every rank then raises, and the launcher sees N non-zero exits, not N - 1 hung ranks)
            failure = None
            try:
                for dr, _, _ in batches:
                    _lib.check(L.gs_tag_phase1_launch(table._h, dr._h))
            except Exception as e:  # noqa: BLE001
                failure = e
            flush_gathers()
            summ_local = []
            try:
                if failure is None:
                    for dr, _, _ in batches:
                        has, lastloc = C.c_uint32(0), C.c_uint32(0)
                        _lib.check(L.gs_tag_phase1_wait(table._h, dr._h, C.byref(has), C.byref(lastloc)))
                        summ_local.append((has.value, lastloc.value))
            except Exception as e:  # noqa: BLE001
                failure = e
            mine = next(((1, l) for h, l in reversed(summ_local) if h), (0, _lib.GS_NONE))
            summ = gdist.checked_summaries(comm, *mine, error=failure)
            carry = gdist.resolve_carry(summ, rank)
            for (dr, _, _), (h, l) in zip(batches, summ_local):
                _lib.check(L.gs_tag_phase2(table._h, dr._h, carry))
                if h:
                    carry = l
            ctr = _lib.GsCounters()
            for dr, r0, _ in batches:
                n_hits = C.c_uint64(0)
                _lib.check(L.gs_tag_finish(table._h, dr._h, None, 0, C.byref(n_hits), None, C.byref(ctr)))
                c = ctr.as_dict()
                for k_, v in (("rows", n_hits.value), ("raw", c["raw_hits"]), ("emitted", c["emitted"]), ("windows", c["windows"]), ("tiles", c["tiles"]), ("fallback", c["fallback_tiles"])):
                    tot[k_] += v
                # every rank's rows are on its own host now (as at N = 1); the concatenated list is gathered into the root's
                # device memory over xGMI (north_star's "hit buffers concatenated via an RCCL gather") and stays there
                pending_gathers.append((dr, n_hits.value, read_lo + r0))
            last.update(tot)
            if gather_to_host:
                gathered["rows"] = flush_gathers(to_host=True)
            return None

        pending_gathers = []

        def flush_gathers(to_host=False):
            rows_g = []
            while pending_gathers:
                dr, n_hits, off = pending_gathers.pop(0)
                g = comm.gather_hits(table, dr, n_hits, off, to_host=to_host)
                if to_host and g is not None:
                    rows_g.append(g.copy())
            return rows_g

        # global index of this rank's first read (the ranks before it hold shards of the same size)
        read_lo = rank * sum(len(b[2]) for b in batches)

        def sync():
            drain_single()  # (every batch's rows are on the host: the pipeline is empty)
            if world > 1:
                comm.barrier()
            _lib.check(L.gs_device_synchronize(device))

        for i in range(warmup):
            step()
            drain_single()
            log(f"{wname} warmup {i}: {last['rows']} rows")
        if world > 1:
            flush_gathers()  # (the timed region starts with nothing pending)
        L.gs_counters_reset(table._h)
        sync()
        step_s = []
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            step()
            step_s.append(time.perf_counter() - t1)
        if world > 1:
            flush_gathers()  # (the last pass's rows: inside the timed region)
        sync()
        dt = time.perf_counter() - t0
        ctr = _lib.GsCounters()
        # (the table's counters accumulate probe / pass times over every launch of the timed steps)
        n_launch = steps * nb
        if world > 1:
            dt = comm.all_max(dt)
            total_bases = int(comm.all_sum(n_bases))
        else:
            total_bases = n_bases
        log(f"{wname}: {steps} steps: {total_bases * steps / dt / 1e9:.1f} Gbase/s")
        # one more pass outside the timed region: the head batch's rows for the parity leg / the gathered list on the root
        census = None
        if hasattr(L, "gs_tuning_census"):  # (libgavisunk_tuning.so, GAVISUNK_LIB=...: what the region fetches of one pass find)
            L.gs_tuning_census(device, None, 1)
        head = step(keep_head=True, gather_to_host=world > 1)
        if hasattr(L, "gs_tuning_census"):
            cz = (C.c_uint64 * 16)()
            L.gs_tuning_census(device, cz, 0)
            names = ["steps", "runs", "runs_with_hit", "runs_deferred_only", "runs_empty_region", "runs_mmer_absent", "runs_hot_region",
                     "windows_deferred", "windows_hit", "windows"]
            census = {n_: int(cz[i]) for i, n_ in enumerate(names)}
            log(f"{wname} census of one pass: {census}")
        gather_check = None
        if world > 1:
            tot_rows = int(comm.all_sum(int(last["rows"])))
            if rank == 0:
                g = gathered["rows"]
                gather_check = bool(g is not None and sum(len(x) for x in g) == tot_rows)
                log(f"gathered list: {sum(len(x) for x in g) if g is not None else None} rows of {tot_rows}: {gather_check}")
        L.gs_table_counters(table._h, C.byref(ctr))
        solo = None
        if world > 1:
            # the reference point for the scaling of THIS workload: the same ranks, the same resident reads, no exchange and no gather
            keep = dict(last)
            step_single()
            sync()
            t2 = time.perf_counter()
            for _ in range(2):
                step_single()
            sync()
            solo = comm.all_max(time.perf_counter() - t2) / 2
            last.update(keep)
            log(f"{wname}: without exchange / gather: {total_bases / solo / 1e9:.1f} Gbase/s")
        res = {"w": w, "wname": wname, "shards": list(shards), "table": table, "batches": batches, "n_bases": n_bases, "total_bases": total_bases, "dt": dt,
               "steps": steps, "step_s": step_s, "n_sunks": n_sunks, "n_groups": n_groups, "sample": sample, "head": head,
               "last": dict(last), "rlen": rlen_all, "n_launch": n_launch, "gather_check": gather_check, "ctr": ctr.as_dict(), "stream": stream, "solo": solo, "census": census, "kept_sunks": kept_sunks}
        return res

    def describe(res, headline):
        """The JSON fragment of one workload run (rank 0)."""
        w, table, last = res["w"], res["table"], res["last"]
        n_bases, steps = res["n_bases"], res["steps"]
        c = res["ctr"]
        h_raw = last["raw"] / max(last["windows"], 1)
        h_emit = last["emitted"] / max(last["windows"], 1)
        launches = max(c["launches"], 1)
        probe_launch_s = c["probe_ms"] / launches / 1e3           # average duration of ONE probe-kernel launch (HIP events)
        probe_step_s = probe_launch_s * len(res["batches"])       # all launches of one pass
        pass_step_s = c["total_ms"] / launches / 1e3 * len(res["batches"])
        bpb = alg_bytes_per_base(h_raw, h_emit)
        achieved = n_bases * bpb / probe_step_s / 1e9
        full_genome = a.genome_mb is None and a.reads is None
        default_size = full_genome and not a.group_len
        wkey = f"{res['wname']}:k{K}:{'full' if default_size else f'group_len_{a.group_len}' if full_genome else 'reduced'}"
        traffic, tnote = traffic_for(wkey) if world == 1 else (None, None)
        n50 = synth_gpu.n50(res["rlen"]) if len(res["rlen"]) else 0
        out = {
            "value": res["total_bases"] * steps / res["dt"],
            "unit": "bases/s",
            "ms_per_step": res["dt"] / steps * 1e3,
            "ms_per_step_spread": {"min": min(res["step_s"]) * 1e3, "median": float(np.median(res["step_s"])) * 1e3, "max": max(res["step_s"]) * 1e3,
                                   "note": "N = 1: the streaming engine is not drained between passes (the last three batches of pass i finish while the first of "
                                           "pass i + 1 are submitted, as consecutive blocks of one read file do); the pipeline is drained after the last timed step, "
                                           "inside the timed region: ms_per_step = that whole time / steps, the spread = host-side submission times"},
            **({"no_exchange": {"value": res["total_bases"] / res["solo"], "unit": "bases/s", "ms_per_step": res["solo"] * 1e3,
                                "note": "the same ranks over the same resident reads WITHOUT the carry exchange and the hit gather (every rank's batches "
                                        "through the one-GPU streaming engine, max over ranks): value / this = what the exchange + gather cost; "
                                        "this / n_gpus = the one-GPU rate of this workload"}} if res.get("solo") else {}),
            "config": {"workload": (w["title"] if full_genome else "REDUCED rehearsal (not the headline workload) of " + w["title"]) +
                                   f": {(a.genome_mb if a.genome_mb is not None else w['genome_mb']):g} Mb{' per haplotype' if w['kind'] == 'diploid' else ''} in 24 contigs, exact SUNK set, "
                                   f"{len(res['rlen'])} synthetic ONT reads per GPU, N50 {n50}, k={K}",
                       "reads_per_gpu": int(len(res["rlen"])), "bases_per_gpu": n_bases, "device_batches_per_gpu": len(res["batches"]),
                       "read_shards_per_gpu": len(res["shards"]),
                       "sunks": int(res["n_sunks"]), "sunk_groups": int(res["n_groups"]),
                       "windows": last["windows"], "raw_hits": last["raw"], "emitted_rows": last["emitted"],
                       "h_raw": round(h_raw, 5), "h_emit": round(h_emit, 7), "tiles": last["tiles"],
                       "fallback_tiles": last["fallback"], "table_GiB": round(table.nbytes / 2**30, 2), "table_layout": table.layout,
                       "parallelism": f"read-sharded x{world}, table replicated", "data_plane": data_plane,
                       "gathered_list_checked": res["gather_check"], "env_knobs": knobs,
                       **({"census_of_one_pass": res["census"]} if res.get("census") else {})},
            "roofline": {"bound": "hbm", "kernel": kernel_names(table.layout), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": tnote,
                         "basis": "algorithmic bytes (SURVEY.md 8d: 8.25 B/base + 16 B/raw hit + 12 B/emitted row) of ONE launch / average HIP-event time of one probe-kernel launch "
                                  "(a pass = `launches_per_pass` launches, one per resident device batch); `traffic` = measured HBM bytes per launch (PMC), null when not measured for this workload and source",
                         "launches_per_pass": len(res["batches"]),
                         "alg_bytes_per_launch": int(n_bases * bpb / len(res["batches"])), "alg_bytes_per_base": round(bpb, 3),
                         "kernel_ms": round(probe_launch_s * 1e3, 4), "kernel_ms_per_pass": round(probe_step_s * 1e3, 4),
                         "pass_ms": round(pass_step_s * 1e3, 4), "kernel_bases_per_s": n_bases / probe_step_s},
        }
        return out

    def release(res):
        if res.get("kept_sunks"):
            res["kept_sunks"][0].close()
        if res.get("stream") is not None:
            res["stream"].close()
        for dr, _, _ in res["batches"]:
            dr.close()
        res["table"].close()

    # The headline workload is the same at every N: config 3 (BASELINE.json configs[2]), every rank its own 1 000 000 reads with
    # the table replicated -- weak scaling, so that value(N) / (N x value(1)) is an efficiency.  BASELINE.json configs[3]
    # (the diploid read set cut over the ranks: strong scaling) is the extra key at N > 1, one of its shards the extra key at N = 1.
    wname = a.workload if a.workload != "auto" else "config3"
    if wname == "diploid":
        per_rank = a.shards if a.shards else (max(8 // world, 1) if world > 1 else 1)
        shards = list(range(rank * per_rank, (rank + 1) * per_rank))
        bps = a.batches or 8
    else:
        shards = [rank]
        bps = a.batches or 8
    res = run_workload(wname, shards, bps, a.steps, a.warmup, want_sample=True,
                       want_chain=((world == 1 and not a.no_chain) or (world > 1 and not a.no_sharded_chain)))

    out = None
    if rank == 0:
        d = describe(res, True)
        out = {"metric": f"ONT bases/sec SUNK-tagged (k={K}), bit-exact hit list", "value": d["value"], "unit": "bases/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": d["ms_per_step"], "ms_per_step_spread": d["ms_per_step_spread"],
               "higher_is_better": True,
               # diploid at N > 1: the FIXED ~1.2e11-base read set (8 shards) cut over the ranks = strong scaling; otherwise every
               # rank tags its own shard of the same size = weak scaling
               "scaling": "strong" if (wname == "diploid" and world > 1 and not a.shards) else "weak",
               "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": d["config"], "roofline": d["roofline"]}
        if "no_exchange" in d:
            out["no_exchange"] = d["no_exchange"]
        # ---- parity on the head of the stream + CPU baseline (outside the timed region) --------------------------
        sample = res["sample"]
        oracle_rows = None
        if sample is not None:
            from oracle import oracle

            table = res["table"]
            seqs, sub = sample["seqs"], sample["sub"]
            nchk = len(seqs)
            otable = oracle.Table(K, sub["code"], sub["code"], (sub["pos"] >> np.uint64(32)).astype(np.uint32), sub["group"].astype(np.int64))
            buf, off = oracle.concat_reads(seqs)
            orc, oh, _, octr = otable.tag(buf, off, cap=1 << 24)
            oracle_rows = {"read": oh["read"], "pos": oh["pos"], "row": sub["rows"][oh["row"]]}
            # (1) the same reads through the engine on their own (all counters comparable) ...
            chk = engine.DeviceReads.from_ascii(seqs, K, device)
            st, gh, _, gctr = engine.tag(table, chk)
            chk.close()
            ok1 = st == 0 and orc == 0 and same_hits(gh, oh, sub["rows"]) and gctr["raw_hits"] == octr["raw_hits"] and gctr["windows"] == octr["windows"]
            # (2) ... and the head of the big pass's own hit list (the first resident batch, fetched after the timed steps)
            if world == 1 and res["head"] is not None:
                big = res["head"]
                ok2 = same_hits(big[big["read"] < nchk], oh, sub["rows"])
            else:
                ok2 = None
            out["parity"] = {"checked_reads": nchk, "checked_bases": int(off[-1]), "rows": int(len(oh)), "raw_hits": int(octr["raw_hits"]),
                             "bit_exact": bool(ok1 and ok2 is not False), "head_of_timed_launch_bit_exact": ok2,
                             "oracle": "oracle/gs_oracle.c over the sample with the sub-table of SUNKs that occur in it"}
            if not out["parity"]["bit_exact"]:
                out["value"] = 0.0  # a fast wrong answer is not a result
            threads = min(16, os.cpu_count() or 1)
            rate, n, reps, bases, secs = timed_oracle(otable, seqs, threads, a.cpu_seconds)
            out["cpu_baseline"] = {"value": rate, "unit": "bases/s", "cores": threads, "kind": "port",
                                   "sample": f"first {n} reads of the rank-0 stream x {reps} passes ({bases} bases) as {threads} parallel chunks, {secs:.1f} s wall; "
                                             f"oracle/gs_oracle.c, one thread per chunk (Snakemake-scatter style); its hash table holds only the {len(sub['rows'])} "
                                             f"SUNKs that occur in the sample ({len(sub['rows']) * 100.0 / max(res['n_sunks'], 1):.2f} % of the set): cache-friendlier than the reference's full table, i.e. a CPU-favourable baseline",
                                   "per_core": rate / threads,
                                   "calibration": {"reference_binary_bases_per_s_per_core": 4.79e6, "where": "workflow/scripts/kmerpos_annot3 (the reference's Nim ELF) timed in the build container on config 2, tagging phase only (SURVEY.md section 6; the binary cannot travel to the GPU box)",
                                                   "port_over_reference_per_core": rate / threads / 4.79e6}}
        if res.get("kept_sunks") and world == 1:
            # the device-resident second half over this workload's own hit list (the rows of the last pass are still in HBM)
            try:
                lr = engine.LocRows.from_sunkset(res["kept_sunks"][0])
                out["pair_stage"] = chain_leg(res["table"], res["batches"], lr, res["kept_sunks"][1], device)
                out["pair_stage"]["workload"] = "the headline workload's own hit list (one pass over every resident batch)"
                log(f"pair stage: {out['pair_stage']['rows_in']} rows in {out['pair_stage']['seconds_total']:.3f} s = {out['pair_stage']['rows_per_s'] / 1e6:.1f} Mrow/s")
                both = both_halves_leg(res["table"], res["batches"], lr, res["kept_sunks"][1], device, max(3, min(a.steps, 10)), 2, res["n_bases"])
                both["consistent_with_pair_stage"] = bool(all(both["summary_of_last_pass"][k_] == out["pair_stage"][k_] for k_ in
                                                              ("rows_in", "reads_with_a_best_contig", "rows_after_diag_filter", "tsv_rows", "validated_intervals")))
                both["tagger_alone"] = out["value"]
                out["engine_both_halves"] = both
                try:
                    out["cli_config3_rows"] = rows_text_leg(res["table"], res["batches"], lr, res["kept_sunks"][0], res["kept_sunks"][1], device,
                                                            oracle_rows)
                    log(f"rows as text: {out['cli_config3_rows']['rows']} rows, {out['cli_config3_rows']['bytes'] / 2**30:.2f} GiB in {out['cli_config3_rows']['seconds']:.2f} s")
                except Exception as e:  # (a measurement leg must not cost the headline)
                    out["cli_config3_rows"] = {"failed": repr(e)}
                log(f"both halves: {both['value'] / 1e9:.1f} Gbase/s ({both['ms_per_step']:.1f} ms per step; the tagger alone {out['value'] / 1e9:.1f})")
                lr.close()
            except Exception as e:  # (a measurement leg must not cost the headline)
                out["pair_stage"] = {"failed": repr(e)}
        if not a.no_engine and world == 1:
            try:
                out.update(engine_leg(a, res["table"], res["batches"], device))
            except Exception as e:  # (a measurement leg must not cost the headline)
                out["engine_note"] = f"engine leg failed: {e}"
    # ---- N > 1: the read-sharded second half, AFTER everything the headline line needs has been measured.  Its collectives
    # (all-reduce of the reads-per-SUNK counters, variable-size gather of the TSV rows) have never run on more than one GPU before
    # the driver's scaling run: whatever happens here -- an exception on some rank, a collective that never returns -- the line
    # still goes out, once, with what was measured before it, and every rank exits 0 (a watchdog THREAD: a main thread stuck in a
    # C call never runs a signal handler) ----
    line_written = threading.Lock()

    def emit_line_and_exit(why):
        if rank == 0 and line_written.acquire(blocking=False):
            out["pair_stage"] = {"failed": why, "workload": "every rank's own hit list of the headline workload (read-sharded second half)"}
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        os._exit(0)

    if world > 1 and not a.no_sharded_chain and res.get("kept_sunks"):
        watchdog = threading.Timer(a.sharded_chain_timeout, emit_line_and_exit, args=(f"gave up after {a.sharded_chain_timeout:g} s",))
        watchdog.daemon = True
        watchdog.start()
        try:
            lr = engine.LocRows.from_sunkset(res["kept_sunks"][0])
            sharded = sharded_chain_leg(comm, res["table"], res["batches"], lr, res["kept_sunks"][1], device, rank * len(res["rlen"]))
            lr.close()
        except BaseException as e:  # noqa: BLE001 (this rank leaves the run here: the others' watchdogs let them follow)
            watchdog.cancel()
            log(f"sharded pair stage failed on rank {rank}: {e!r}")
            emit_line_and_exit(repr(e))
        watchdog.cancel()
        if rank == 0 and sharded is not None:
            out["pair_stage"] = sharded
            out["pair_stage"]["workload"] = "every rank's own hit list of the headline workload (read-sharded second half)"
            log(f"sharded pair stage: {sharded['rows_in']} rows on {world} ranks in {sharded['seconds_total']:.2f} s = {sharded['rows_per_s'] / 1e6:.1f} Mrow/s")
    release(res)
    # ---- the other genome-scale workload as an extra key --------------------------------------------------------------
    if not a.no_extra and a.workload == "auto" and K == 20 and ((a.genome_mb is None and a.reads is None) or a.rehearse_extra) and not a.group_len:
        if world == 1:
            ex = run_workload("diploid", [0], 8, max(3, a.steps // 2), 1, want_sample=False)
            if rank == 0:
                d = describe(ex, False)
                out["diploid_shard"] = {k_: d[k_] for k_ in ("value", "unit", "ms_per_step", "config", "roofline")}
                out["diploid_shard"]["note"] = "BASELINE.json configs[3] on ONE GPU: the shard (1/8 of the reads) a rank of an 8-GPU run tags; the N > 1 headline lines divide this workload's 8 shards over the ranks"
            release(ex)
        else:
            per_rank = max(8 // world, 1)
            ex = run_workload("diploid", list(range(rank * per_rank, (rank + 1) * per_rank)), 8, 3, 1, want_sample=False)
            if rank == 0:
                d = describe(ex, False)
                out["diploid_strong"] = {k_: d[k_] for k_ in ("value", "unit", "ms_per_step", "config", "roofline", "no_exchange") if k_ in d}
                out["diploid_strong"]["scaling"] = "strong"
                out["diploid_strong"]["note"] = ("BASELINE.json configs[3]: the two-haplotype table replicated, the FIXED 40x read set (8 shards of ~15 Gbase, "
                                                 "~1.2e11 bases) cut over the ranks -- total work fixed, so this key scales strongly; divide by the N = 1 line's diploid_shard.value x 8 shards")
            release(ex)
    # ---- every other BASELINE.json config, and the realistic group length, driver-timed (a few steps each) -------------
    if not a.no_extra and a.workload == "auto" and K == 20 and a.genome_mb is None and a.reads is None and not a.group_len and world == 1:
        for key, wname, k_, gl, note in (
                ("config5_k21", "config5", 21, 0, "BASELINE.json configs[4] at k = 21 (config/config.yaml:2 sets the SUNK length)"),
                ("config5_k25", "config5", 25, 0, "BASELINE.json configs[4] at k = 25"),
                ("config3_group_len_200", "config3", 20, 200,
                 "the headline genome with a 44-base interspersed element every 200 bases of unique sequence: location groups of ~157 SUNKs (7.75e6 groups) instead of "
                 "~12 000 -- workflow/rules/defineSUNKs.smk:124-126 makes a group per merged interval, a real T2T haplotype sits between this and the headline")):
            try:
                K, a.group_len = k_, gl
                ex = run_workload(wname, [0], 8, 3, 1, want_sample=False)
                d = describe(ex, False)
                out[key] = {k2: d[k2] for k2 in ("value", "unit", "ms_per_step", "config", "roofline")}
                out[key]["steps"] = 3
                out[key]["note"] = note
                release(ex)
            except Exception as e:  # (a measurement leg must not cost the headline)
                out[key] = {"failed": repr(e)}
            finally:
                K, a.group_len = 20, 0
    if rank == 0:
        if not a.no_config2 and world == 1 and K == 20:
            out["config2"] = config2_leg(a, device)
            if a.cli and not a.no_cli_large and a.workload == "auto" and a.genome_mb is None:
                try:
                    out["cli_1e8_keys"] = cli_large_leg(a, device)
                    if "cli_genome_scale" in out["cli_1e8_keys"]:
                        out["cli_genome_scale"] = out["cli_1e8_keys"].pop("cli_genome_scale")
                except Exception as e:  # (a measurement leg must not cost the headline)
                    out["cli_1e8_keys"] = {"failed": repr(e)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
