"""RCCL data plane of a read-sharded run, through the C-ABI (gs_comm_*): table broadcast and hit gather.
The unique id travels over the caller's control plane (dist.Comm: torch.distributed / gloo)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine
from ._lib import check, lib

ID_BYTES = 128


class RcclComm:
    def __init__(self, ctrl, device: int):
        """ctrl: an object with rank, world and broadcast_bytes(bytes, src) (dist.Comm)."""
        L = lib()
        buf = C.create_string_buffer(ID_BYTES)
        if ctrl.rank == 0:
            check(L.gs_comm_unique_id(buf))
        idb = ctrl.broadcast_bytes(buf.raw, src=0)
        h = C.c_void_p()
        check(L.gs_comm_init(device, ctrl.world, ctrl.rank, idb, C.byref(h)))
        self._h, self.rank, self.world, self.device = h, ctrl.rank, ctrl.world, device

    def broadcast_table(self, table, root: int = 0):
        """The root's built table on every rank (device-to-device over xGMI); returns this rank's table."""
        out = C.c_void_p()
        check(lib().gs_table_broadcast(self._h, root, table._h if table is not None else None, C.byref(out)))
        if self.rank == root:
            return table
        t = engine.DeviceTable.__new__(engine.DeviceTable)
        t._h, t.device = out, self.device
        t.k = t.n = None
        return t

    def allgather_pairs(self, a: int, b: int):
        """ncclAllGather of two 64-bit words per rank (gs_comm_allgather_u64) -> [(a, b)] in rank order."""
        mine = np.array([a, b], dtype=np.uint64)
        out = np.zeros(2 * self.world, dtype=np.uint64)
        check(lib().gs_comm_allgather_u64(self._h, mine.ctypes.data, 2, out.ctypes.data))
        return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(self.world)]

    def gather_hits(self, table, reads, counts, read_offset: int, dst: int = 0, to_host: bool = True):
        """All ranks' hit lists of the last pass concatenated on rank dst: in its device memory, and (to_host) as a numpy
        HIT_DTYPE view of pinned host memory; None on the other ranks and when the rows are left on the device."""
        n = C.c_uint64(0)
        check(lib().gs_gather_hits(self._h, dst, table._h, reads._h, read_offset, None, 0, C.byref(n)))
        self.gathered_rows = n.value
        if self.rank != dst or not to_host:
            return None
        vp = C.c_void_p()
        check(lib().gs_gather_view(self._h, C.byref(vp), None))  # pinned host memory of the library: no further copy
        return engine._view(vp.value, n.value, engine.HIT_DTYPE)

    def reducer(self):
        """(function pointer, context) of gs_comm_reduce_u32: what engine.RowSet.bad_keys(reduce=) takes -- one ncclAllReduce of the
        reads-per-SUNK counters per haplotype."""
        return C.cast(lib().gs_comm_reduce_u32, C.c_void_p), self._h

    def gatherv(self, data: np.ndarray, status: int = 0, dst: int = 0):
        """gs_comm_gatherv_host: every rank's bytes on rank dst, as a list of uint8 arrays in rank order (None elsewhere); a rank
        that passes a non-zero status makes every rank raise that error."""
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        sizes = np.zeros(self.world, np.uint64)
        out = C.c_void_p()
        check(lib().gs_comm_gatherv_host(self._h, dst, data.ctypes.data if len(data) else None, len(data), status, C.byref(out), sizes.ctypes.data))
        if self.rank != dst:
            return None
        try:
            total = int(sizes.sum())
            blob = engine._view(out.value, total, np.uint8).copy() if total else np.zeros(0, np.uint8)
        finally:
            lib().gs_host_free(out)
        cuts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        return [blob[cuts[r]:cuts[r + 1]] for r in range(self.world)]

    def close(self):
        if getattr(self, "_h", None):
            lib().gs_comm_destroy(self._h)
            self._h = None

    __del__ = close
