#!/usr/bin/env python3
"""Headline benchmark: ONT bases/s SUNK-tagged (k=20) on N MI355X, bit-exact hit list.

Workload = BASELINE.json configs[1] ("config 2"): one 5 Mb contig with
pasted repeat families, its exact SUNK set (k=20), 10 000 synthetic ONT reads (N50 ~ 30 kb, 6 %
errors, both strands) PER GPU (weak scaling: each rank tags its own shard of one logical input
file; the SUNK table is built once on rank 0 and replicated through the collective backend).

A step = one full tagging pass over the rank's resident reads (inputs already in HBM): probe
kernel, tile scans, emit kernel, hit list to the host; at N > 1 also the carry exchange
(all_gather) and the RCCL gather of the hit buffers to rank 0.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` (probe kernel,
algorithmic bytes / HIP-event time, against 8 TB/s HBM) and `cpu_baseline` (the oracle's C
restatement of the reference tagger timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
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


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def alg_bytes_per_base(h_raw: float, h_emit: float) -> float:
    """0.25 B (2-bit read stream) + 8 B (one key-slot probe per window)
    + 16 B per raw hit (payload fetch) + 12 B per emitted hit (output record)."""
    return 8.25 + 16.0 * h_raw + 12.0 * h_emit


def cpu_baseline(k, sunks_code, sunks_group, seqs, threads, target_seconds=20.0):
    """The oracle (oracle/gs_oracle.c, `port` of kmerpos_annot3.nim's main loop) on this box's
    host cores: `threads` independent chunks, like Snakemake's scatter of the reference."""
    from oracle import oracle

    table = oracle.Table(k, sunks_code, sunks_code, np.zeros(len(sunks_code), np.uint32), sunks_group)
    # calibrate on a few reads (twice: the first pass pays the page faults of the table)
    buf, off = oracle.concat_reads(seqs[:32])
    for _ in range(2):
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
            table.tag(packed[i][0], packed[i][1], cap=1 << 22)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    bases = int(sum(int(p[1][-1]) for p in packed)) * reps
    return {"value": bases / dt, "unit": "bases/s", "cores": threads, "kind": "port",
            "sample": f"first {n} reads of the rank-0 shard x {reps} passes ({bases} bases) as {threads} parallel chunks, "
                      f"{dt:.1f} s wall; oracle/gs_oracle.c, one thread per chunk (Snakemake-scatter style)"}, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=10000, help="reads per GPU (config 2: 10000)")
    ap.add_argument("--seed", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-time budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split-groups", type=int, default=0,
                    help="experiment, not the headline workload: cut location groups every N bases (more, smaller groups)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    from gavisunk_amd import _lib, engine, synth

    L = _lib.lib()
    if L.gs_device_count() < 1:
        sys.exit("bench.py: no HIP device (gavisunk_amd has no CPU fallback)")
    device = local_rank % L.gs_device_count()

    dist = None
    tdev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        from gavisunk_amd import dist as gdist

        backend = os.environ.get("GAVISUNK_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N > 1 path on one GPU
        if backend == "nccl":
            torch.cuda.set_device(device)
            tdev = torch.device("cuda", device)
            dist.init_process_group("nccl", device_id=tdev)
        else:
            dist.init_process_group(backend)

    k = 20
    # ---- inputs (synthetic; SURVEY.md Appendix B recipe) --------------------------------
    t0 = time.perf_counter()
    genome, rng = synth.genome_config2(a.seed)
    if rank == 0:
        sunks = synth.define_sunks([("ctg5M", synth.codes_to_ascii(genome))], k)
        group_key = sunks.group if not a.split_groups else sunks.group * (1 << 24) + (sunks.start - sunks.group) // a.split_groups
        _, loc = np.unique(group_key, return_inverse=True)
        keys, loc, payload, group = sunks.code, loc.astype(np.uint32), np.arange(len(sunks), dtype=np.uint32), sunks.group
    else:
        keys = loc = payload = group = None
    if world > 1:
        keys = gdist.broadcast_array(keys, np.uint64, tdev)
        loc = gdist.broadcast_array(loc, np.uint32, tdev)
        payload = gdist.broadcast_array(payload, np.uint32, tdev)
    log(f"genome + {len(keys)} SUNKs in {time.perf_counter() - t0:.1f} s")
    t0 = time.perf_counter()
    rrng = np.random.default_rng([a.seed, rank])  # every rank simulates its own shard of reads
    lens = synth.lognormal_lengths(rrng, a.reads, median=19000.0)
    reads = synth.simulate_reads(rrng, [genome], lens)
    seqs = [synth.codes_to_ascii(r) for r in reads]
    del reads
    n_bases = int(sum(len(s) for s in seqs))
    srt = np.sort(lens)[::-1]
    n50 = int(srt[np.searchsorted(np.cumsum(srt), srt.sum() / 2)])
    log(f"{a.reads} reads, {n_bases} bases, N50 {n50} in {time.perf_counter() - t0:.1f} s")

    t0 = time.perf_counter()
    table = engine.DeviceTable(k, keys, loc, payload, device)
    batch = engine.ReadBatch.from_ascii(seqs)
    t_pack = time.perf_counter() - t0
    t0 = time.perf_counter()
    dev_reads = engine.DeviceReads(batch, k, device)
    t_up = time.perf_counter() - t0
    log(f"table {table.nbytes / 2**20:.0f} MiB; pack {t_pack:.2f} s, upload {t_up:.2f} s "
        f"({batch.n_words * 8 / 2**20:.0f} MiB packed)")

    hit_buf = None
    if world > 1:
        import torch
        hit_buf = torch.empty((max(dev_reads.n_windows // 64, 1 << 20), 3), dtype=torch.int32, device=tdev)
        on_gpu = tdev is not None
    n_hits = C.c_uint64(0)
    carry = C.c_uint32(0)
    ctr = _lib.GsCounters()
    host_hits = np.zeros(1 << 20, dtype=engine.HIT_DTYPE)
    read_lo = rank * a.reads

    def step():
        if world == 1:
            _lib.check(L.gs_tag_launch(table._h, dev_reads._h, _lib.GS_NONE))
            st = L.gs_tag_finish(table._h, dev_reads._h, host_hits.ctypes.data, len(host_hits), C.byref(n_hits),
                                 C.byref(carry), C.byref(ctr))
            _lib.check(st)
            return None
        has, last = C.c_uint32(0), C.c_uint32(0)
        _lib.check(L.gs_tag_phase1(table._h, dev_reads._h, C.byref(has), C.byref(last)))
        summ = gdist.exchange_summaries(has.value, last.value, tdev)
        _lib.check(L.gs_tag_phase2(table._h, dev_reads._h, gdist.resolve_carry(summ, rank)))
        _lib.check(L.gs_tag_finish(table._h, dev_reads._h, None, 0, C.byref(n_hits), C.byref(carry), C.byref(ctr)))
        copy = L.gs_tag_hits_to_device if on_gpu else L.gs_tag_hits_to_host
        _lib.check(copy(table._h, dev_reads._h, hit_buf.data_ptr(), hit_buf.shape[0], None))
        return gdist.gather_hits(hit_buf[: n_hits.value], read_lo)

    def sync():
        if world > 1:
            import torch
            dist.barrier()
            if tdev is not None:
                torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    L.gs_counters_reset(table._h)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        gathered = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([n_bases], dtype=torch.int64, device=tdev)
        dist.all_reduce(tot)
        total_bases = int(tot.item())
    else:
        total_bases = n_bases

    if rank == 0:
        c = ctr.as_dict()
        h_raw = c["raw_hits"] / max(c["windows"], 1)
        h_emit = c["emitted"] / max(c["windows"], 1)
        probe_s = c["probe_ms"] / max(c["launches"], 1) / 1e3
        pass_s = c["total_ms"] / max(c["launches"], 1) / 1e3
        bpb = alg_bytes_per_base(h_raw, h_emit)
        achieved = n_bases * bpb / probe_s / 1e9
        # HBM-side bytes per launch come from a separate rocprofv3 --pmc run (never collected inside the bench);
        # they are quoted only for the exact workload they were measured on
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01", "traffic.json")))
            if a.reads == 10000 and a.seed == 20 and world == 1 and not a.split_groups:
                traffic = tj[table.layout.split(":")[0]]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "ONT bases/sec SUNK-tagged (k=20), bit-exact hit list",
            "value": total_bases * a.steps / dt,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "config 2 (BASELINE.json configs[1]): single 5 Mb contig, "
                                   f"{a.reads} synthetic ONT reads per GPU, N50 {n50}, k=20",
                       "reads_per_gpu": a.reads, "bases_per_gpu": n_bases, "sunks": int(len(keys)),
                       "windows": c["windows"], "raw_hits": c["raw_hits"], "emitted_rows": c["emitted"],
                       "h_raw": round(h_raw, 5), "h_emit": round(h_emit, 7), "tiles": c["tiles"],
                       "table_MiB": round(table.nbytes / 2**20, 1), "table_layout": table.layout,
                       "parallelism": f"read-sharded x{world}, table replicated"},
            "roofline": {"bound": "hbm", "kernel": {"compact-lds": "k_join_bin+k_join_segT+k_join_probe+k_join_merge",
                                    "compact-l2": "k_probe_part+k_merge_part+k_probe"}.get(table.layout.split(":")[0], "k_probe"), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_note": "HBM bytes per launch from profiles/r01/traffic.json (PMC run, gfx950-corrected)" if traffic else None,
                         "alg_bytes_per_launch": int(n_bases * bpb),
                         "alg_bytes_per_base": round(bpb, 3), "kernel_ms": round(probe_s * 1e3, 4),
                         "pass_ms": round(pass_s * 1e3, 4),
                         "kernel_bases_per_s": n_bases / probe_s},
        }
        # ---- parity on the rank-0 shard prefix + CPU baseline (outside the timed region) -------
        if not a.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            cb, otable = cpu_baseline(k, keys, group, seqs, threads, a.cpu_seconds)
            out["cpu_baseline"] = cb
            from oracle import oracle
            nchk = min(len(seqs), 300)
            buf, off = oracle.concat_reads(seqs[:nchk])
            orc, oh, _, octr = otable.tag(buf, off, cap=1 << 22)
            chk = engine.DeviceReads(engine.ReadBatch.from_ascii(seqs[:nchk]), k, device)
            st, gh, _, gctr = engine.tag(table, chk)
            ok = (st == 0 and orc == 0 and len(gh) == len(oh) and np.array_equal(gh["read"], oh["read"])
                  and np.array_equal(gh["pos"], oh["pos"]) and np.array_equal(gh["payload"], oh["row"])
                  and gctr["raw_hits"] == octr["raw_hits"])
            out["parity"] = {"checked_reads": nchk, "rows": int(len(oh)), "bit_exact": bool(ok)}
            if not ok:
                out["value"] = 0.0  # a fast wrong answer is not a result
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
