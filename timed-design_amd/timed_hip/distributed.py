"""Multi-GPU prediction: contiguous frame shards, one gather.

The reference is single-process (its only parallelism is a multiprocessing.Pool in the sampler),
so nothing here replaces reference code; it is what lets rank 0 write the per-chain ``.csv`` /
``.fasta`` exactly as reference predict.py:161-185 does when the frames of one dataset were
predicted on several GPUs.

Frames are independent at inference (BatchNorm uses moving statistics, no cross-frame op), so the
flat dataset map — whose order is fixed by create_flat_dataset_map (reference utils.py:362-393) — is
cut into ``world`` contiguous ranges ``[floor(i*N/W), floor((i+1)*N/W))``; every rank holds the full
(small) weights, predicts its range, and the ``[n_i, n_classes]`` fp32 blocks are gathered to rank 0
in rank order, which IS map order.  One process per GPU (``torch.distributed.run`` sets
RANK/LOCAL_RANK/WORLD_SIZE).

Two transports for the single exchange step:
  * ``RcclGather``  device buffers, RCCL grouped send/recv over xGMI through the C ABI
                    (th_comm_gather_rows) — the production path on a GPU node.  Its control plane (the
                    128-byte RCCL id, "did every rank come up", text sizes) is ``timed_hip.rendezvous``:
                    plain TCP sockets, NO PyTorch anywhere in the product's N > 1 path;
  * any object with the same three methods over host arrays — the CPU suite drives the real predict.py control flow
                    on 2 and 8 gloo ranks with ``tests/_gloo_transport.GlooGather`` (torch.distributed lives in tests/
                    only: nothing in this package imports PyTorch).
Both present ``gather_rows(local, counts, root) -> ndarray | None`` plus the small control-plane calls
``allgather_ints(values) -> [[...] per rank]`` and ``barrier()``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous ranges [floor(i*n/world), floor((i+1)*n/world)); sizes differ by at most one."""
    if world < 1 or n < 0:
        raise ValueError("need world >= 1 and n >= 0")
    return [((i * n) // world, ((i + 1) * n) // world) for i in range(world)]


def shard_counts(n: int, world: int) -> List[int]:
    return [hi - lo for lo, hi in shard_bounds(n, world)]


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class RcclGather:
    """Row gather of DEVICE buffers through th_comm_* (RCCL over xGMI).  ``unique_id`` is created on
    rank 0 with ``RcclGather.new_unique_id()`` and shipped to the other ranks out of band (e.g. a gloo
    broadcast, see bench.py)."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device: int, rendezvous=None):
        self._lib = _lib.load()
        self.rank, self.world, self.device = rank, world, device
        if world > 1 and rendezvous is None:
            raise ValueError("an RcclGather over more than one rank needs its HostRendezvous (use from_environment)")
        self.rendezvous = rendezvous
        self._h = C.c_void_p()
        _lib.check(self._lib.th_comm_init(unique_id, world, rank, device, C.byref(self._h)))

    @classmethod
    def from_environment(cls, rank: int, world: int, device: int, rendezvous=None) -> "RcclGather":
        """Bring the communicator up inside a one-process-per-GPU job (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as
        ``torch.distributed.run`` or any other launcher sets them): rank 0 creates the RCCL unique id and it travels
        to the other ranks over ``timed_hip.rendezvous`` (TCP; no PyTorch).  Raises on every rank when any rank fails
        — a GPU job never degrades silently to a host exchange."""
        from .rendezvous import HostRendezvous
        own = rendezvous is None
        rdzv = rendezvous or HostRendezvous(rank, world)
        ident, err = bytes(_lib.TH_COMM_ID_BYTES), ""
        if rank == 0:
            try:
                ident = cls.new_unique_id()
            except _lib.TimedHipError as e:
                err = str(e)
        ident = rdzv.broadcast(ident)
        comm = None
        if any(ident):
            try:
                comm = cls(ident, world, rank, device, rendezvous=rdzv)
            except _lib.TimedHipError as e:
                err = str(e)
        if rdzv.all_min(1 if comm is not None else 0) != 1:
            if comm is not None:
                comm.rendezvous = None
                comm.close()
            if own:
                rdzv.close()
            raise RuntimeError(f"RCCL communicator could not be created on every rank (rank {rank}: {err or 'ok'})")
        comm._owns_rendezvous = own
        return comm

    @staticmethod
    def new_unique_id() -> bytes:
        buf = C.create_string_buffer(_lib.TH_COMM_ID_BYTES)
        _lib.check(_lib.load().th_comm_unique_id(buf))
        return buf.raw

    def gather_rows_device(self, d_local: int, counts: Sequence[int], width: int, root: int, d_out: int = 0):
        arr = (C.c_int64 * self.world)(*[int(c) for c in counts])
        _lib.check(self._lib.th_comm_gather_rows(self._h, C.c_void_p(d_local), arr, int(width), int(root),
                                                 C.c_void_p(d_out)))

    def allgather_ints(self, values: Sequence[int]) -> List[List[int]]:
        if self.world == 1:
            return [[int(v) for v in values]]
        return self.rendezvous.allgather_ints(values)

    def barrier(self):
        _lib.check(self._lib.th_comm_barrier(self._h))

    def stats(self) -> dict:
        """What this rank's gathers issued so far: ncclSend / ncclRecv calls and bytes, device copies of the root's own block."""
        out = (C.c_int64 * 5)()
        _lib.check(self._lib.th_comm_stats(self._h, out))
        return dict(sends=out[0], recvs=out[1], bytes_sent=out[2], bytes_received=out[3], root_copies=out[4])

    def close(self):
        if self._h:
            self._lib.th_comm_free(self._h)
            self._h = C.c_void_p()
        if getattr(self, "_owns_rendezvous", False) and self.rendezvous is not None:
            self.rendezvous.close()
            self.rendezvous = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostGather:
    """The same row-gather interface over the product's own TCP rendezvous (timed_hip/rendezvous.py), rows in HOST memory: for
    ranks that SHARE a GPU — a development box, the 2-process-on-1-GPU tests — where RCCL cannot span the ranks (it refuses two
    ranks on one device).  Never chosen silently: predict.py takes it only under TIMED_GATHER=host; a one-process-per-GPU job
    uses RcclGather and fails loudly when RCCL cannot be brought up."""

    def __init__(self, rank: int, world: int, rendezvous=None):
        from .rendezvous import HostRendezvous
        self.rank, self.world = rank, world
        self._own = rendezvous is None
        self.rendezvous = rendezvous or HostRendezvous(rank, world)

    def gather_rows(self, local: np.ndarray, counts: Sequence[int], root: int = 0) -> Optional[np.ndarray]:
        local = np.ascontiguousarray(local, dtype=np.float32)
        if len(counts) != self.world or local.shape[0] != counts[self.rank]:
            raise ValueError(f"rank {self.rank}: local block has {local.shape[0]} rows, counts say {list(counts)}")
        width = local.shape[1]
        parts = self.rendezvous.allgather(local.tobytes())
        if self.rank != root:
            return None
        blocks = [np.frombuffer(b, dtype=np.float32).reshape(c, width) for b, c in zip(parts, counts)]
        return np.concatenate(blocks, axis=0) if blocks else np.empty((0, width), np.float32)

    def allgather_ints(self, values: Sequence[int]) -> List[List[int]]:
        return self.rendezvous.allgather_ints(values)

    def barrier(self):
        self.rendezvous.barrier()

    def close(self):
        if self._own and self.rendezvous is not None:
            self.rendezvous.close()
        self.rendezvous = None


def predict_sharded(model, frames_for_range, n_total: int, gather, root: int = 0) -> Optional[np.ndarray]:
    """Predict this rank's contiguous shard and gather the probability rows to ``root``.

    ``frames_for_range(lo, hi) -> ndarray[hi-lo, D,H,W,C]`` loads the frames of map rows [lo, hi)
    (e.g. ``lambda lo, hi: load_batch(path, flat_map[lo:hi])[0]``); ``gather`` is a host transport (tests/_gloo_transport.GlooGather)
    transport.  Returns the full [n_total, n_classes] matrix in map order on ``root``, None elsewhere.
    """
    counts = shard_counts(n_total, gather.world)
    lo, hi = shard_bounds(n_total, gather.world)[gather.rank]
    if hi > lo:
        local = model.predict(frames_for_range(lo, hi))
    else:
        local = np.empty((0, model.n_classes), dtype=np.float32)
    return gather.gather_rows(local, counts, root)
