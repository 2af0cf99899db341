"""Read-sharded multi-GPU tagging: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference has no distributed runtime; its only parallelism is Snakemake's per-chunk scatter
(workflow/Snakefile:23-24, tagONT.smk:19-37).  Here ONE input file is cut into contiguous
blocks of reads, one block per rank, with the SUNK table replicated.  Reads are independent
except for `prevLoc` (kmerpos_annot3.nim:82-84), which the reference never resets between
reads; so the path has exactly one exchange step:

  phase 1  every rank probes its shard and learns {has a raw hit?, loc of its last raw hit}
  exchange all_gather of that 2-word summary (8 bytes per rank: latency-bound, not link-bound)
  phase 2  every rank resolves its carry-in = last loc of the nearest earlier shard that has a
           raw hit, and finishes its pass
  gather   per-rank hit buffers (12 B per output row) are concatenated on rank 0 in shard order (gs_gather_hits: counts and
           statuses by ncclAllGather, rows by grouped send / receive of exactly their size).

The SECOND half shards the same way ("pair-distance stage: shard by read"): every rank runs the device
chain (csrc/gs_chain.hip) over the rows of its OWN reads -- the hit lists never leave their GPU -- and three things cross ranks:

  capacity   the slot order of diag_filter_v3's Table depends on its capacity, which persists from read to read: one all_gather of
             the capacity each shard leaves (4 bytes), and a shard whose predecessors grew the Table past 64 slots runs again
  counters   the bad-SUNK histogram counts reads per SUNK over ALL reads: one all_reduce (sum) of n_keys x 4 bytes per haplotype
  TSV rows   the pair stage's output rows (20 bytes each, ~1e-2 of the hit list) go to the root, which merges them in name
             order and runs the contig-wide union over all of them (gs_pairstage_from_rows)

Nothing here computes on the CPU: the tagging itself is engine.DeviceTable / engine.DeviceReads, the filters engine.RowSet.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np

from ._lib import GS_NONE

ROW_DTYPE = np.dtype([("contig", "<u4"), ("name", "<u4"), ("rank", "<u4"), ("id", "<i8")], align=False)  # a TSV row on its way to the root


def shard_bounds(n_reads: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of reads of `rank` (sizes differ by at most one read)."""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def resolve_carry(summaries: Sequence[Tuple[int, int]], rank: int, carry0: int = GS_NONE) -> int:
    """Carry-in of `rank` from the (has_raw, last_loc) summaries of all ranks: the last loc of the
    nearest earlier shard with a raw hit, else the carry into the whole file."""
    for r in range(rank - 1, -1, -1):
        has, loc = summaries[r]
        if has:
            return int(loc)
    return int(carry0)


PASS_FAILED = 2  # in the "has a raw hit" word of a phase-1 summary: this rank's pass failed before the exchange


class RankFailed(RuntimeError):
    """A rank of a read-sharded run failed before an exchange; raised on EVERY rank after it."""


def checked_summaries(comm, has_raw: int, last_loc: int, error: BaseException | None = None):
    """The one exchange of a read-sharded pass with a status in it: a rank whose phase 1 failed (`error`) still takes part -- its
    `has a raw hit` word says PASS_FAILED -- and every rank raises RankFailed after the exchange, so that all of them leave the run
    (exit code != 0) instead of waiting in the next collective for a rank that is gone.  -> the summaries of all ranks."""
    summ = comm.exchange_summaries(PASS_FAILED if error is not None else has_raw, last_loc)
    failed = [r for r, (h, _) in enumerate(summ) if h == PASS_FAILED]
    if failed:
        raise RankFailed(f"the tagging pass of rank(s) {failed} failed before the carry exchange" + (f": {error}" if error is not None else "")) from error
    return summ


class Comm:
    """One rank of a read-sharded run.  Control plane = torch.distributed over gloo (rendezvous, barrier, the 8-byte phase-1
    summaries, scalars); data plane = RCCL behind the C-ABI (gs_comm_*: table broadcast, hit gather) when `rccl`.
    torch never touches the GPU here: a torch wheel bundles its own HIP runtime, and two HIP runtimes cannot both hold the
    device in one process, so all device work stays in libgavisunk.so."""

    def __init__(self, device: int, rccl: bool = True):
        import torch.distributed as dist

        if not dist.is_initialized():
            dist.init_process_group("gloo")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.rccl = None
        if rccl:
            from . import rccl as _rccl

            self.rccl = _rccl.RcclComm(self, device)

    # ---- control plane -------------------------------------------------------------------------------------
    def barrier(self):
        self.dist.barrier()

    def _all_gather_i64(self, vals):
        import torch

        mine = torch.tensor([int(v) for v in vals], dtype=torch.int64)
        out = torch.empty((self.world, len(vals)), dtype=torch.int64)
        self.dist.all_gather_into_tensor(out, mine.reshape(1, -1))
        return out.tolist()

    def exchange_summaries(self, has_raw: int, last_loc: int):
        """{has a raw hit, loc of the last one} of every rank: ncclAllGather on the library's stream when the data plane is
        RCCL, the gloo control plane otherwise."""
        if self.rccl is not None:
            return self.rccl.allgather_pairs(has_raw, last_loc)
        return [(int(h), int(l)) for h, l in self._all_gather_i64([has_raw, last_loc])]

    def all_max(self, x: float) -> float:
        import torch

        t = torch.tensor([float(x)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(self, x: int) -> int:
        return sum(r[0] for r in self._all_gather_i64([x]))

    def broadcast_bytes(self, data, src: int = 0) -> bytes:
        import torch

        n = torch.tensor([len(data) if self.rank == src else 0], dtype=torch.int64)
        self.dist.broadcast(n, src=src)
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8) if self.rank == src else torch.empty(int(n.item()), dtype=torch.uint8)
        self.dist.broadcast(t, src=src)
        return t.numpy().tobytes()

    # ---- data plane ----------------------------------------------------------------------------------------
    def gather_hits(self, table, reads, n_local: int, read_offset: int, dst: int = 0, to_host: bool = True):
        """Per-rank hit lists of the last pass concatenated on rank `dst` in shard order (numpy HIT_DTYPE), None elsewhere.
        to_host=False (RCCL data plane only): the concatenated rows stay in dst's device memory (every rank's own rows already
        are on its own host)."""
        from . import engine

        if self.rccl is not None:
            return self.rccl.gather_hits(table, reads, None, read_offset, dst, to_host=to_host)
        import torch

        from ._lib import GS_OK, GsError, lib

        # every rank takes part in the exchange of the counts, which carries its status (a KeyError is data-dependent: one shard
        # meets it, kmerpos_annot3.nim:90): all ranks then raise the same error instead of waiting for each other
        mine = np.zeros(max(n_local, 1), dtype=engine.HIT_DTYPE)
        st = lib().gs_tag_hits_to_host(table._h, reads._h, mine.ctypes.data, len(mine), None) if n_local else \
            lib().gs_tag_hits_to_host(table._h, reads._h, None, 0, None)
        meta = self._all_gather_i64([n_local if st == GS_OK else 0, st])
        for r, (_, code) in enumerate(meta):
            if code != GS_OK:
                raise GsError(int(code), f"the pass of rank {r} failed" + (": key not found [KeyError]" if code == -2 else ""))
        counts = [m[0] for m in meta]
        if n_local:
            mine["read"][:n_local] += np.uint32(read_offset)
        width = max(max(counts), 1)
        padded = torch.zeros((width, 3), dtype=torch.int32)
        padded[:n_local] = torch.from_numpy(mine[:n_local].view(np.int32).reshape(-1, 3))
        if self.rank == dst:
            bufs = [torch.empty_like(padded) for _ in range(self.world)]
            self.dist.gather(padded, bufs, dst=dst)
            cat = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0).numpy()
            return np.ascontiguousarray(cat).view(engine.HIT_DTYPE).reshape(-1)
        self.dist.gather(padded, None, dst=dst)
        return None

    # ---- the two exchanges of the second half -----------------------------------------------------------------
    def reducer(self):
        """(function pointer, context) for engine.RowSet.bad_keys(reduce=): the reads-per-SUNK counters of all ranks summed in
        place -- ncclAllReduce inside the library on the RCCL data plane; device -> host -> gloo all_reduce -> device otherwise."""
        if self.rccl is not None:
            return self.rccl.reducer()
        import torch

        from ._lib import lib

        def cb(_ctx, d_ptr, n):
            try:
                host = np.zeros(max(int(n), 1), np.uint32)
                if lib().gs_device_copy_to_host(self.device, host.ctypes.data, d_ptr, int(n) * 4):
                    return -1
                t = torch.from_numpy(host.astype(np.int64))
                self.dist.all_reduce(t)
                host[:] = t.numpy().astype(np.uint32)
                return lib().gs_device_copy_from_host(self.device, d_ptr, host.ctypes.data, int(n) * 4)
            except Exception:  # (an exception cannot cross the C frame)
                return -1

        self._reduce_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)(cb)  # (kept alive by the communicator)
        return C.cast(self._reduce_cb, C.c_void_p), None

    def gatherv(self, data: np.ndarray, status: int = 0, dst: int = 0):
        """Every rank's bytes on rank dst as a list of uint8 arrays in rank order (None elsewhere); a non-zero status on any rank
        makes every rank raise.  RCCL: gs_comm_gatherv_host; gloo: sizes first, then a padded gather."""
        if self.rccl is not None:
            return self.rccl.gatherv(data, status, dst)
        import torch

        from ._lib import GsError

        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        meta = self._all_gather_i64([0 if status else len(data), status])
        for r, (_, code) in enumerate(meta):
            if code:
                raise GsError(int(code), f"the step of rank {r} failed")
        sizes = [m[0] for m in meta]
        padded = torch.zeros(max(max(sizes), 1), dtype=torch.uint8)
        padded[: len(data)] = torch.from_numpy(data.copy())
        if self.rank == dst:
            bufs = [torch.empty_like(padded) for _ in range(self.world)]
            self.dist.gather(padded, bufs, dst=dst)
            return [b[:c].numpy().copy() for b, c in zip(bufs, sizes)]
        self.dist.gather(padded, None, dst=dst)
        return None

    def close(self):
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None
        self.dist.barrier()
        self.dist.destroy_process_group()


def carried_capacity(caps: Sequence[int], rank: int) -> int:
    """The capacity of diag_filter_v3's Table before the first read of `rank`: the largest any earlier shard left (64 = fresh)."""
    return max([64] + [int(c) for c in caps[:rank]])


def sharded_second_half(comm: Comm, rows, contig_names, in_fai, key_hap, name_grank, name_long, read_offset: int, other_hap=None,
                        dst: int = 0):
    """Diagonal filter -> step 2 -> bad SUNKs -> process-by-contig (pair test, component) over THIS rank's rows on its GPU, the
    contig-wide union on rank dst (module docstring).  rows: engine.RowSet of this rank's reads, name ids = read index in the shard;
    name_grank[n_local]: position of each of them in the sorted order of ALL read names (`groupby("rname")`);
    name_long[n_local]: the read is >= 10 kb; read_offset: index of the shard's first read in the whole input; other_hap: the
    other haplotype's row set of the same reads (bad SUNKs count both; default: `rows` stands for both, as bench.py has it).
    -> (local, merged): local = this rank's diag verdicts and bad-SUNK flags; merged (rank dst only, else None) = the dict
    engine.RowSet.pair_stage returns on one GPU over all reads, `row_name` = index of the read in the whole input."""
    from . import engine
    from ._lib import GsError

    status, err = 0, None
    diag = None
    cap = 64
    try:
        *diag, kept, cap = rows.diag_filter(contig_names, in_fai, want_cap=True)
    except GsError as e:
        status, err = e.status, e
    meta = comm.exchange_summaries(cap, status & 0xFFFFFFFF)  # (the capacity travels with the status: a failed rank fails all)
    for r, (_, code) in enumerate(meta):
        if code:
            raise err if err is not None else GsError(int(code) - (1 << 32), f"the diagonal filter of rank {r} failed")
    cap_in = carried_capacity([m[0] for m in meta], comm.rank)
    # every step between two exchanges runs under the same rule: a rank that fails keeps walking the SAME sequence of collectives as
    # the others, carrying its status, and all ranks raise together at the next agreement -- nobody waits inside a collective for a
    # rank that has left
    try:
        if cap_in > 64:  # (a read with more than 43 distinct contigs somewhere before this shard: rare, and exact either way)
            *diag, kept, _ = rows.diag_filter(contig_names, in_fai, cap_in=cap_in, want_cap=True)
        rows.keep_filtered()
    except GsError as e:
        status, err = e.status, e
    meta = comm.exchange_summaries(0, status & 0xFFFFFFFF)  # agreed BEFORE the counters' all-reduce
    for r, (_, code) in enumerate(meta):
        if code:
            raise err if err is not None else GsError(int(code) - (1 << 32), f"step 2 of rank {r} failed")
    bad = meancov = d_bad = None
    try:  # (gs_rowset_bad_keys_reduced enters the reduce once per haplotype whatever happens locally)
        bad, meancov, d_bad = engine.RowSet.bad_keys(rows, other_hap if other_hap is not None else rows, key_hap, keep_on_device=True,
                                                     reduce=comm.reducer())
    except GsError as e:
        status, err = e.status, e
    res = None
    try:
        if not status:
            res = rows.pair_stage(name_grank, name_long, d_key_bad=d_bad)
    except GsError as e:
        status, err = e.status, e
    finally:
        if d_bad:
            engine.free_device(d_bad, rows.device)
    # this rank's TSV rows (read index in the whole input, global name rank) and its input contigs, in one message
    t = len(res["row_id"]) if res is not None else 0
    rec = np.zeros(t, ROW_DTYPE)
    ci = np.zeros(0, np.uint32)
    if t or res is not None:
        ci = res["contigs_in"].astype(np.uint32)
        rec["contig"], rec["name"], rec["id"] = res["row_contig"], res["row_name"] + np.uint32(read_offset), res["row_id"]
        rec["rank"] = np.asarray(name_grank, np.uint32)[res["row_name"]]
    blob = np.concatenate([np.array([len(ci)], np.uint32).view(np.uint8), ci.view(np.uint8), rec.view(np.uint8).reshape(-1)])
    parts = comm.gatherv(blob, status, dst)
    if err is not None:
        raise err
    local = {"diag": diag, "rows_kept": kept, "bad": bad, "meancov": meancov, "cap_in": cap_in, "counters": res["counters"]}
    if parts is None:
        return local, None
    cis, recs = [], []
    for b in parts:
        nci = int(b[:4].view(np.uint32)[0])
        cis.append(b[4:4 + 4 * nci].view(np.uint32))
        recs.append(b[4 + 4 * nci:].view(ROW_DTYPE))
    allrec = np.concatenate(recs)
    order = np.lexsort((allrec["rank"], allrec["contig"]))  # stable: a read's rows stay in the order its rank wrote them
    allrec = allrec[order]
    merged = engine.pairstage_from_rows(allrec["contig"], allrec["name"], allrec["id"], np.unique(np.concatenate(cis)), rows.device)
    return local, merged
