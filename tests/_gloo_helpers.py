"""torch.distributed (gloo) stand-ins for the library's data plane, for the CPU protocol tests only (the product's collectives
are RCCL calls inside libgavisunk.so: gs_comm_allgather_u64, gs_table_broadcast, gs_gather_hits)."""
from typing import List, Optional, Tuple

import numpy as np


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
