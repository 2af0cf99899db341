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

Nothing here computes on the CPU: the tagging itself is engine.DeviceTable / engine.DeviceReads.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from ._lib import GS_NONE


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

    def close(self):
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None
        self.dist.barrier()
        self.dist.destroy_process_group()
