"""Multi-GPU plumbing for the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* kNN queries are independent → replicas + sharded query batches need NO data-path collective.
* A sharded index (config 3: rowid mod world → one HNSW graph per GPU) has exactly one exchange step:
  every rank searches the same queries on its shard, then the per-shard top-k lists — k x (int64 id,
  f32 distance) per query — are all-gathered and merged.  merge order is total and deterministic:
  (distance, shard rank, position in the shard's list).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_of_rowid(rowid: int, world: int) -> int:
    return int(rowid) % world


def allgather_merge_topk(ids: torch.Tensor, dists: torch.Tensor, counts: torch.Tensor, k: int, group=None):
    """ids [nq,k] int64 (-1 padded), dists [nq,k] f32, counts [nq] int32 — this rank's per-shard results
    (ascending by distance).  Returns the merged global top-k (ids, dists, counts) on every rank."""
    world = dist.get_world_size(group)
    nq = ids.shape[0]
    gi = [torch.empty_like(ids) for _ in range(world)]
    gd = [torch.empty_like(dists) for _ in range(world)]
    gc = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(gi, ids.contiguous(), group=group)
    dist.all_gather(gd, dists.contiguous(), group=group)
    dist.all_gather(gc, counts.contiguous(), group=group)
    ai = torch.stack(gi, 1).reshape(nq, world * k)            # [nq, world*k], shard-major
    ad = torch.stack(gd, 1).reshape(nq, world * k)
    ac = torch.stack(gc, 1)                                    # [nq, world]
    pos = torch.arange(k, device=ids.device).repeat(world).unsqueeze(0)
    valid = pos < ac.repeat_interleave(k, dim=1)
    key = torch.where(valid, ad, torch.full_like(ad, float("inf")))
    # stable sort keeps (shard rank, position) order among equal distances
    order = torch.sort(key, dim=1, stable=True).indices[:, :k]
    oi = torch.gather(ai, 1, order)
    od = torch.gather(ad, 1, order)
    ov = torch.gather(valid, 1, order)
    oi = torch.where(ov, oi, torch.full_like(oi, -1))
    od = torch.where(ov, od, torch.zeros_like(od))
    return oi, od, ov.sum(1).to(counts.dtype)


def max_over_ranks(seconds: float, device) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
