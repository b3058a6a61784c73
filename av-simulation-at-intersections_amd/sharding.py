"""Multi-GPU: egos are independent (no term of the QP couples two egos, main/lib/mpc.py:141-211), so the
batch shards across ranks with NO collective on the solve path.  One process per GPU; the only exchange
is the final gather of the controls/trajectories (RCCL all-gather over xGMI when the backend is "nccl",
gloo on CPU for tests)."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np
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



class CabiGather:
    """The same gather through the C-ABI (jsim_comm_unique_id / jsim_comm_init / jsim_mpc_gather: RCCL's ncclAllGather called by
    libjsim_mpc.so itself, no torch collective).  torch.distributed -- any backend -- is only the side channel that carries the
    128-byte ncclUniqueId from rank 0 to the other ranks once; a caller with its own rendezvous passes `unique_id` instead."""

    def __init__(self, engine, rank: int = None, world: int = None, unique_id: bytes = None, group=None):
        from . import _cabi
        self.eng, self._cabi = engine, _cabi
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank, self.world = int(rank), int(world)
        lib = engine.lib
        if unique_id is None:
            buf = (C.c_char * 128)()
            if self.rank == 0:
                _cabi.check(lib.jsim_comm_unique_id(C.cast(buf, C.c_void_p)), None, "jsim_comm_unique_id")
            if self.world > 1:
                t = torch.from_numpy(np.frombuffer(bytes(buf), dtype=np.uint8).copy())
                if dist.get_backend(group) == "nccl":
                    t = t.to(engine.device)
                dist.broadcast(t, src=0, group=group)
                unique_id = bytes(t.cpu().numpy())
            else:
                unique_id = bytes(buf)
        if len(unique_id) != 128:
            raise ValueError("an ncclUniqueId is 128 bytes")
        idbuf = C.create_string_buffer(unique_id, 128)
        _cabi.check(lib.jsim_comm_init(engine._ctx, C.cast(idbuf, C.c_void_p), self.world, self.rank), engine._ctx, "jsim_comm_init")

    def gather_rows(self, local: torch.Tensor, B: int) -> torch.Tensor:
        """All-gather per-ego rows [b_local, ...] into [B, ...] in rank order (ragged shards are padded to the largest)."""
        sizes = shard_sizes(B, self.world)
        per = max(sizes)
        if local.shape[0] != sizes[self.rank]:
            raise ValueError("local shard has the wrong number of rows")
        if not local.is_cuda:
            raise ValueError("jsim_mpc_gather moves device memory")
        tail = tuple(local.shape[1:])
        if local.shape[0] < per:
            local = torch.cat([local, torch.zeros((per - local.shape[0],) + tail, dtype=local.dtype, device=local.device)], dim=0)
        local = local.contiguous()
        out = torch.empty((self.world * per,) + tail, dtype=local.dtype, device=local.device)
        nbytes = local.numel() * local.element_size()
        self._cabi.check(self.eng.lib.jsim_mpc_gather(self.eng._ctx, None, C.c_void_p(local.data_ptr()), C.c_void_p(out.data_ptr()),
                                                      nbytes, self.eng._stream()), self.eng._ctx, "jsim_mpc_gather")
        if all(sz == per for sz in sizes):
            return out
        return torch.cat([out[r * per: r * per + sizes[r]] for r in range(self.world)], dim=0)

    def close(self):
        self._cabi.check(self.eng.lib.jsim_comm_destroy(self.eng._ctx), self.eng._ctx, "jsim_comm_destroy")
