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
into the root's device memory.

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
    return {"compact-lds": "k_join_bin+k_join_segT+k_join_probe+k_join_merge", "compact-l2": "k_probe_part+k_merge_part+k_probe",
            "minimizer-hbm": "k_probe_mz"}.get(layout.split(":")[0], "k_probe")


def source_hash() -> str:
    """Hash of the kernel sources as the compiler sees them: `//` comments and blank lines do not count (so that a comment
    edit does not orphan the measured traffic figure; any code edit does)."""
    h = hashlib.sha256()
    for f in ("gs_device.hip", "gs_mz.h"):
        for ln in open(os.path.join(ROOT, "gavisunk_amd", "csrc", f), "r"):
            code = ln.split("//", 1)[0].rstrip()
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def traffic_for(workload_key: str):
    """HBM bytes per launch from a separate rocprofv3 --pmc run (never collected inside the bench); quoted only for the
    exact workload and kernel sources they were measured on (profiles/r02/traffic.json)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r02", "traffic.json")))
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
# config 3: everything on the device
# ------------------------------------------------------------------------------------------------------------------
def config3(a, device, rank, comm):
    """-> table, packed reads, read lengths, SUNK count, group count, oracle sample, the read set (device ASCII).
    With an RCCL communicator only rank 0 defines the SUNKs and builds the table; the others receive the built table."""
    from gavisunk_amd import engine, synth_gpu

    t0 = time.perf_counter()
    gen = synth_gpu.Genome(int(a.genome_mb * 1e6), n_contigs=24, seed=a.seed, device=device)
    log(f"assembly surrogate: {gen.n_bases} bases in 24 contigs ({time.perf_counter() - t0:.1f} s)")
    replicate = comm is not None and comm.rccl is not None
    sunks = table = None
    if rank == 0 or not replicate:
        t0 = time.perf_counter()
        sunks = engine.DeviceSunkSet(gen.ptr, gen.off, K, device)
        log(f"exact SUNK set: {sunks.n} keys in {sunks.n_groups} groups ({time.perf_counter() - t0:.1f} s, gs_sunks_define_device)")
        t0 = time.perf_counter()
        table = sunks.table()
        log(f"table: {table.layout} | {table.nbytes / 2**30:.1f} GiB ({time.perf_counter() - t0:.1f} s)")
    if replicate:
        t0 = time.perf_counter()
        table = comm.rccl.broadcast_table(table, root=0)
        log(f"table replicated over RCCL ({time.perf_counter() - t0:.1f} s, gs_table_broadcast)")
    t0 = time.perf_counter()
    rng = np.random.default_rng([a.seed, rank])
    lens = synth_gpu.lognormal_lengths(rng, a.reads, median=32000.0)
    rs = synth_gpu.ReadSet(gen, lens, seed=a.seed * 1000 + rank)
    rlen = (rs.off[1:] - rs.off[:-1]).astype(np.int64)
    log(f"{a.reads} reads, {rs.n_bases} bases, N50 {synth_gpu.n50(rlen)} ({time.perf_counter() - t0:.1f} s)")
    gen.close()
    t0 = time.perf_counter()
    dev_reads = engine.device_reads_from_device_ascii(rs.ptr, rs.off, K, device)
    log(f"packed on the device: {dev_reads.n_windows} windows ({time.perf_counter() - t0:.1f} s)")
    # ---- the sample the oracle walks: the first reads of the stream, and the SUNKs that can matter for them ----
    sample = None
    if rank == 0 and not a.no_cpu_baseline:
        t0 = time.perf_counter()
        from oracle import oracle

        ns = min(a.check_reads, a.reads)
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
    n_sunks, n_groups = (sunks.n, sunks.n_groups) if sunks is not None else (0, 0)
    if sunks is not None:
        sunks.close()
    return table, dev_reads, rlen, n_sunks, n_groups, sample, rs


# ------------------------------------------------------------------------------------------------------------------
# config 2 (BASELINE.json configs[1]): the pinned inputs of tests/golden/config2_full
# ------------------------------------------------------------------------------------------------------------------
def config2_leg(a, device):
    from gavisunk_amd import _lib, engine, synth
    from oracle import oracle

    L = _lib.lib()
    t0 = time.perf_counter()
    seq, sunks, names, reads = synth.config2(seed=20, n_reads=10000, k=K)
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
        res = {}
        for tag, path in (("fasta", f"{d}/reads.fa"),):
            t0 = time.perf_counter()
            rc = kmerpos_annot3.main([path, f"{d}/db", f"{d}/loc", f"{d}/out.sunkpos"])
            dt = time.perf_counter() - t0
            ok = rc == 0 and open(f"{d}/out.sunkpos", "rb").read().split(b"\n")[:-1] == want
            res[tag] = {"seconds": round(dt, 3), "cli_bases_per_s": n_bases / dt, "bit_exact": bool(ok)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (config 3: 1 000 000)")
    ap.add_argument("--genome-mb", type=float, default=3100.0, help="assembly size in Mb (config 3: 3100)")
    ap.add_argument("--seed", type=int, default=30)
    ap.add_argument("--k", type=int, default=20, help="SUNK length (headline: 20; 21 / 25 = BASELINE.json configs[4] rehearsals)")
    ap.add_argument("--check-reads", type=int, default=1000, help="reads of the stream the oracle walks for the parity leg")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-time budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (parity + CPU baseline)")
    ap.add_argument("--no-config2", action="store_true", help="skip the config-2 extra key")
    ap.add_argument("--cli", action="store_true", help="(default now) config-2 leg: also time the command-line drop-in (.fa -> .sunkpos)")
    ap.add_argument("--no-cli", action="store_true", help="config-2 leg without the command-line timing")
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
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    # tuning / layout knobs change what is measured: refuse them (GAVISUNK_DEVICE and the library override are harmless)
    knobs = sorted(v for v in os.environ if v.startswith("GAVISUNK_") and v not in ("GAVISUNK_DEVICE", "GAVISUNK_BENCH_BACKEND"))
    if knobs and not os.environ.get("GAVISUNK_BENCH_ALLOW_KNOBS"):
        sys.exit(f"bench.py: unset {knobs} (tuning knobs change the measured path); GAVISUNK_BENCH_ALLOW_KNOBS=1 for experiments")

    from gavisunk_amd import _lib, engine

    L = _lib.lib()
    if L.gs_device_count() < 1:
        sys.exit("bench.py: no HIP device (gavisunk_amd has no CPU fallback)")
    device = local_rank % L.gs_device_count()

    comm = None
    data_plane = "single GPU"
    if world > 1:
        from gavisunk_amd import dist as gdist

        # control plane: torch.distributed over gloo (rendezvous, barrier, 8-byte summaries); data plane: RCCL behind the
        # C-ABI (gs_comm_*).  torch never touches the GPU in this process: its wheel bundles a second HIP runtime.
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
                data_plane = "RCCL over xGMI (gs_table_broadcast, gs_gather_hits)"

    table, dev_reads, rlen, n_sunks, n_groups, sample, rs = config3(a, device, rank, comm)
    n_bases = int(rlen.sum())
    from gavisunk_amd import synth_gpu

    n50 = synth_gpu.n50(rlen)
    if world > 1 and rank != 0:
        sample = None

    n_hits = C.c_uint64(0)
    carry = C.c_uint32(0)
    ctr = _lib.GsCounters()
    host_hits = np.zeros(1, dtype=engine.HIT_DTYPE)
    read_lo = rank * a.reads
    full_reads = dev_reads

    def step():
        nonlocal host_hits
        if world == 1:
            # the emit kernel writes the rows to pinned host memory itself: the pass ends with the hit list on the host
            _lib.check(L.gs_tag_launch(table._h, dev_reads._h, _lib.GS_NONE))
            _lib.check(L.gs_tag_finish(table._h, dev_reads._h, None, 0, C.byref(n_hits), C.byref(carry), C.byref(ctr)))
            vp = C.c_void_p()
            _lib.check(L.gs_tag_hits_view(table._h, dev_reads._h, C.byref(vp), None))
            host_hits = engine._view(vp.value, n_hits.value, engine.HIT_DTYPE)
            return None
        has, last = C.c_uint32(0), C.c_uint32(0)
        _lib.check(L.gs_tag_phase1(table._h, dev_reads._h, C.byref(has), C.byref(last)))
        summ = comm.exchange_summaries(has.value, last.value)
        _lib.check(L.gs_tag_phase2(table._h, dev_reads._h, gdist.resolve_carry(summ, rank)))
        _lib.check(L.gs_tag_finish(table._h, dev_reads._h, None, 0, C.byref(n_hits), C.byref(carry), C.byref(ctr)))
        # every rank's rows are on its own host now (gs_tag_finish, as at N = 1); the concatenated list is gathered into the
        # root's device memory over xGMI (north_star's "hit buffers concatenated via an RCCL gather") and stays there
        return comm.gather_hits(table, dev_reads, n_hits.value, read_lo, to_host=gather_to_host)

    gather_to_host = False

    def sync():
        if world > 1:
            comm.barrier()
        _lib.check(L.gs_device_synchronize(device))

    for i in range(a.warmup):
        step()
        log(f"warmup {i}: {n_hits.value} rows")
    L.gs_counters_reset(table._h)
    sync()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t1 = time.perf_counter()
        step()
        step_s.append(time.perf_counter() - t1)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = comm.all_max(dt)
        total_bases = int(comm.all_sum(n_bases))
    else:
        total_bases = n_bases
    log(f"{a.steps} steps: {total_bases * a.steps / dt / 1e9:.1f} Gbase/s")
    gather_check = None
    if world > 1:  # once, outside the timed region: the gathered list on the root's host = every rank's rows, rank 0's first
        gather_to_host = True
        g = step()
        gather_to_host = False
        tot = int(comm.all_sum(int(n_hits.value)))
        if rank == 0:
            vp = C.c_void_p()
            _lib.check(L.gs_tag_hits_view(table._h, dev_reads._h, C.byref(vp), None))
            own = engine._view(vp.value, n_hits.value, engine.HIT_DTYPE)
            gather_check = bool(g is not None and len(g) == tot and np.array_equal(g[: len(own)], own)
                                and (len(g) == len(own) or int(g["read"][len(own)]) >= a.reads))
            log(f"gathered list: {len(g) if g is not None else None} rows of {tot}, head = rank 0's rows: {gather_check}")
    c_main = ctr.as_dict()  # counters of the timed passes (the strong-scaling leg below runs more)
    # strong scaling beside the weak headline (N > 1): the SAME total read count as one GPU tags, cut over the ranks
    strong = None
    if world > 1:
        per = max(a.reads // world, 1)
        dev_reads = engine.device_reads_from_device_ascii(rs.ptr, rs.off[: per + 1], K, device)
        read_lo = rank * per
        step()
        sync()
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        sync()
        sdt = comm.all_max(time.perf_counter() - t1)
        sbases = int(comm.all_sum(int(rs.off[per])))
        strong = {"total_reads": per * world, "total_bases": sbases, "steps": 3, "ms_per_step": sdt / 3 * 1e3, "value": sbases * 3 / sdt,
                  "unit": "bases/s", "note": "fixed total work (the one-GPU read count) cut over the ranks; the headline above is weak scaling"}
        dev_reads.close()
        dev_reads = full_reads
        read_lo = rank * a.reads
    rs.close()

    if rank == 0:
        c = c_main
        h_raw = c["raw_hits"] / max(c["windows"], 1)
        h_emit = c["emitted"] / max(c["windows"], 1)
        probe_s = c["probe_ms"] / max(c["launches"], 1) / 1e3
        pass_s = c["total_ms"] / max(c["launches"], 1) / 1e3
        bpb = alg_bytes_per_base(h_raw, h_emit)
        achieved = n_bases * bpb / probe_s / 1e9
        full = a.genome_mb == 3100.0 and a.reads == 1_000_000 and K == 20
        wkey = f"config3:{a.genome_mb:g}Mb:{a.reads}reads:seed{a.seed}"
        traffic, tnote = traffic_for(wkey) if world == 1 else (None, None)
        out = {
            "metric": f"ONT bases/sec SUNK-tagged (k={K}), bit-exact hit list",
            "value": total_bases * a.steps / dt,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "ms_per_step_spread": {"min": min(step_s) * 1e3, "median": float(np.median(step_s)) * 1e3, "max": max(step_s) * 1e3},
            "higher_is_better": True,
            "scaling": "weak",
            "strong_scaling": strong,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": ("config 3 (BASELINE.json configs[2]): CHM13-scale haplotype surrogate, " if full else
                                    "REDUCED rehearsal of config 3 (not the headline workload): ") +
                                   f"{a.genome_mb:g} Mb in 24 contigs, exact SUNK set, {a.reads} synthetic ONT reads per GPU, N50 {n50}, k={K}",
                       "reads_per_gpu": a.reads, "bases_per_gpu": n_bases, "sunks": int(n_sunks), "sunk_groups": int(n_groups),
                       "windows": c["windows"], "raw_hits": c["raw_hits"], "emitted_rows": c["emitted"],
                       "h_raw": round(h_raw, 5), "h_emit": round(h_emit, 7), "tiles": c["tiles"],
                       "fallback_tiles": c["fallback_tiles"], "table_GiB": round(table.nbytes / 2**30, 2), "table_layout": table.layout,
                       "parallelism": f"read-sharded x{world}, table replicated", "data_plane": data_plane,
                       "gathered_list_checked": gather_check,
                       "env_knobs": knobs},
            "roofline": {"bound": "hbm", "kernel": kernel_names(table.layout), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": tnote,
                         "basis": "algorithmic bytes (SURVEY.md 8d: 8.25 B/base + 16 B/raw hit + 12 B/emitted row) / HIP-event time of the probe kernel; `traffic` = measured HBM bytes per launch (PMC), null when not measured for this workload and source",
                         "alg_bytes_per_launch": int(n_bases * bpb), "alg_bytes_per_base": round(bpb, 3),
                         "kernel_ms": round(probe_s * 1e3, 4), "pass_ms": round(pass_s * 1e3, 4),
                         "kernel_bases_per_s": n_bases / probe_s},
        }
        # ---- parity on the head of the stream + CPU baseline (outside the timed region) --------------------------
        if sample is not None:
            from oracle import oracle

            seqs, sub = sample["seqs"], sample["sub"]
            nchk = len(seqs)
            otable = oracle.Table(K, sub["code"], sub["code"], (sub["pos"] >> np.uint64(32)).astype(np.uint32), sub["group"].astype(np.int64))
            buf, off = oracle.concat_reads(seqs)
            orc, oh, _, octr = otable.tag(buf, off, cap=1 << 24)
            # (1) the same reads through the engine on their own (all counters comparable) ...
            chk = engine.DeviceReads.from_ascii(seqs, K, device)
            st, gh, _, gctr = engine.tag(table, chk)
            chk.close()
            ok1 = st == 0 and orc == 0 and same_hits(gh, oh, sub["rows"]) and gctr["raw_hits"] == octr["raw_hits"] and gctr["windows"] == octr["windows"]
            # (2) ... and the head of the big launch's own hit list
            if world == 1:
                big = host_hits[: n_hits.value]
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
                                             f"SUNKs that occur in the sample ({len(sub['rows']) * 100.0 / max(n_sunks, 1):.2f} % of the set): cache-friendlier than the reference's full table, i.e. a CPU-favourable baseline"}
            # engine rate: host-packed words (pageable host memory) -> hits on host, upload included
            # (the sample four times over: a batch of a few hundred Mbase, so that the fixed cost of a new device batch -- its
            # per-tile arrays, the first launch -- does not dominate; config 2's engine rate below is on its whole read set)
            erep = 4
            batch = engine.ReadBatch.from_ascii(seqs * erep)
            t1 = time.perf_counter()
            d2 = engine.DeviceReads(batch, K, device)
            engine.tag(table, d2)
            t_engine = time.perf_counter() - t1
            d2.close()
            batch.close()
            out["engine_bases_per_s"] = int(off[-1]) * erep / t_engine
            out["engine_note"] = (f"{nchk * erep} reads ({int(off[-1]) * erep} bases: the parity sample x {erep}): gs_reads_upload of host-packed words "
                                  f"(pageable memory) + gs_tag + hits on host, one call each, first use of this batch, {t_engine * 1e3:.1f} ms")
        if not a.no_config2 and world == 1 and K == 20:
            dev_reads.close()
            table.close()
            out["config2"] = config2_leg(a, device)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
