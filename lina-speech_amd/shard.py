"""Batch sharding of the decode job across the GPUs of a node.

Every utterance row owns its conv/recurrent state and token stream, so the batch is split
into contiguous row ranges, one process per GPU, with NO data-path collective
("replicas").  The only exchange is the final gather of the token ids."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_rows(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) row range of `rank`; sizes differ by at most one (ragged batches)."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_tokens(tokens: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather per-rank token ids [Q, B_rank, n] into [Q, total, n] (rows in rank order).
    Uses torch.distributed (RCCL on GPUs, gloo on CPU); a no-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return tokens
    world = dist.get_world_size(group)
    Q, _, n = tokens.shape
    sizes = [shard_rows(total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(Q, width, n, dtype=tokens.dtype, device=tokens.device)
    pad[:, :tokens.shape[1]] = tokens
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:, :hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=1)
