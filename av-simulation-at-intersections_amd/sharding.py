"""Multi-GPU: egos are independent (no term of the QP couples two egos, main/lib/mpc.py:141-211), so the
batch shards across ranks with NO collective on the solve path.  One process per GPU; the only exchange
is the final gather of the controls/trajectories (RCCL all-gather over xGMI when the backend is "nccl",
gloo on CPU for tests)."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of ceil(B / world) egos per rank (the last ranks may be short or empty)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-B // world)
    lo = min(rank * per, B)
    hi = min(lo + per, B)
    return lo, hi


def shard_sizes(B: int, world: int) -> List[int]:
    return [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]


def gather_rows(local: torch.Tensor, B: int, group=None) -> torch.Tensor:
    """All-gather per-ego rows [b_local, ...] into [B, ...] in rank order.  Equal shards use a single
    all_gather_into_tensor (one RCCL collective); ragged shards are padded to the largest shard."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(B, world)
    per = max(sizes)
    tail = local.shape[1:]
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local shard has the wrong number of rows")
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(tail), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()  # rehearsal on a box without RCCL peers: gloo moves host memory only
    out = torch.empty((world * per,) + tuple(tail), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    out = out.to(dev)
    if all(s == per for s in sizes):
        return out
    return torch.cat([out[r * per: r * per + sizes[r]] for r in range(world)], dim=0)
