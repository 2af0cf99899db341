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
  gather   per-rank hit buffers (12 B per output row) are concatenated on rank 0 in shard order:
           counts first, then one padded all_gather-free `gather`.

Nothing here computes on the CPU: the tagging itself is engine.DeviceTable / engine.DeviceReads.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

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


def exchange_summaries(has_raw: int, last_loc: int, device=None) -> List[Tuple[int, int]]:
    """all_gather of the per-shard phase-1 summary (one collective, one device-to-host copy)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    mine = torch.tensor([int(has_raw), int(last_loc)], dtype=torch.int64, device=device)
    out = torch.empty((world, 2), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine.reshape(1, 2))
    rows = out.cpu().tolist()
    return [(int(h), int(l)) for h, l in rows]


def gather_hits(local_hits, read_offset: int, dst: int = 0):
    """Concatenate per-rank hit buffers on rank `dst` in shard order.

    local_hits: int32/uint32-viewable torch tensor of shape [n, 3] (read, pos, payload) on the
    rank's device; read indices are local to the shard and are shifted by `read_offset`.
    Returns the [N, 3] tensor on rank dst, None elsewhere.  Two collectives: the counts
    (all_gather, 8 B per rank), then one gather padded to the largest count."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = local_hits.device
    n = int(local_hits.shape[0])
    n_local = torch.tensor([[n]], dtype=torch.int64, device=dev)
    counts_t = torch.empty((world, 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts_t, n_local)
    counts = [int(c[0]) for c in counts_t.cpu().tolist()]
    width = max(max(counts), 1)
    padded = torch.zeros((width, 3), dtype=torch.int32, device=dev)
    if n:
        padded[:n] = local_hits.to(torch.int32)
        padded[:n, 0] += int(read_offset)
    if rank == dst:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, bufs, dst=dst)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    dist.gather(padded, None, dst=dst)
    return None


def broadcast_array(arr: Optional[np.ndarray], dtype, device=None, src: int = 0) -> np.ndarray:
    """Replicate a host array from rank `src` through the collective backend (the SUNK table's key /
    loc / payload arrays: built once, broadcast, every rank then builds its own device table)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    n = torch.tensor([len(arr) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    nbytes = int(n.item()) * np.dtype(dtype).itemsize
    if rank == src:
        raw = np.ascontiguousarray(arr, dtype=dtype).view(np.uint8)
        t = torch.from_numpy(raw.copy()).to(device) if device is not None else torch.from_numpy(raw.copy())
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().view(dtype)


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
        counts = [r[0] for r in self._all_gather_i64([n_local])]
        import torch

        from ._lib import check, lib

        mine = np.zeros(max(n_local, 1), dtype=engine.HIT_DTYPE)
        if n_local:
            check(lib().gs_tag_hits_to_host(table._h, reads._h, mine.ctypes.data, n_local, None))
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
