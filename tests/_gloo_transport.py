"""TEST transport for the N > 1 control flow of predict.py: the row gather / int all-gather / barrier interface of
timed_hip.distributed.RcclGather over host arrays and torch.distributed's gloo backend.  Lives under tests/ because the product
package imports no PyTorch (its own N > 1 control plane is timed_hip/rendezvous.py, the data path RCCL through the C ABI)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np


class GlooGather:
    """Row gather of host arrays over an initialised torch.distributed (gloo) group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (init_process_group(backend='gloo'))")
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def gather_rows(self, local: np.ndarray, counts: Sequence[int], root: int = 0) -> Optional[np.ndarray]:
        import torch
        local = np.ascontiguousarray(local, dtype=np.float32)
        if local.shape[0] != counts[self.rank]:
            raise ValueError(f"rank {self.rank}: local block has {local.shape[0]} rows, expected {counts[self.rank]}")
        width = local.shape[1]
        # gloo's gather wants equal sizes: pad every block to the largest shard, trim at the root
        mx = max(counts) if counts else 0
        buf = torch.zeros((mx, width), dtype=torch.float32)
        buf[: local.shape[0]] = torch.from_numpy(local)
        if self.rank == root:
            parts = [torch.empty((mx, width), dtype=torch.float32) for _ in range(self.world)]
            self.dist.gather(buf, parts, dst=root, group=self.group)
            return np.concatenate([p[:c].numpy() for p, c in zip(parts, counts)], axis=0)
        self.dist.gather(buf, None, dst=root, group=self.group)
        return None

    def allgather_ints(self, values: Sequence[int]) -> List[List[int]]:
        import torch
        mine = torch.tensor([int(v) for v in values], dtype=torch.int64)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine, group=self.group)
        return [[int(x) for x in p.tolist()] for p in parts]

    def barrier(self):
        self.dist.barrier(group=self.group)
