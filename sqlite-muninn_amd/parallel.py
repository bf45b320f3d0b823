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
    dev = ids.device
    staged = dist.get_backend(group) == "gloo" and dev.type != "cpu"  # gloo rehearsal on a GPU box: stage via host
    src = [t.cpu().contiguous() if staged else t.contiguous() for t in (ids, dists, counts)]
    gi = [torch.empty_like(src[0]) for _ in range(world)]
    gd = [torch.empty_like(src[1]) for _ in range(world)]
    gc = [torch.empty_like(src[2]) for _ in range(world)]
    dist.all_gather(gi, src[0], group=group)
    dist.all_gather(gd, src[1], group=group)
    dist.all_gather(gc, src[2], group=group)
    if staged:
        gi, gd, gc = [t.to(dev) for t in gi], [t.to(dev) for t in gd], [t.to(dev) for t in gc]
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


def node2vec_train_distributed(off, adj, dim, p=1.0, q=1.0, num_walks=10, walk_length=80, window=5, neg_samples=5,
                               learning_rate=0.025, epochs=1, batch_walks=0, device=0, group=None):
    """Data-parallel Node2Vec (config 4): every rank holds a replica of syn0/syn1neg; the walks of each batch are
    split over the ranks (contiguous slices, rank order = walk order); each rank computes the (centre, target, err)
    samples and the per-position neu1e vectors of its slice on its GPU, both are all-gathered (RCCL over xGMI with backend "nccl"; staged through
    the host with "gloo") and every rank applies the whole batch.  Because the gathered sample order equals the
    single-GPU order, the embeddings are bit-identical to mn_node2vec_train(..., MN_N2V_BATCHED) on one GPU.
    Returns (embeddings [n][dim] float32, stats)."""
    import ctypes as C

    import numpy as np

    from .graph import N2vParams, N2vStats, _glib
    from .hnsw import MuninnHipError

    L = _glib()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    off = np.ascontiguousarray(off, np.int32)
    adj = np.ascontiguousarray(adj if len(adj) else np.zeros(1, np.int32), np.int32)
    n = len(off) - 1
    prm = N2vParams(dim, p, q, num_walks, walk_length, window, neg_samples, learning_rate, epochs, batch_walks)
    S = L.mn_n2v_begin(n, off, adj, C.byref(prm), device)
    if not S:
        raise MuninnHipError((L.mn_node2vec_last_error() or b"").decode())
    try:
        B, cap, pcap = L.mn_n2v_batch_walks(S), L.mn_n2v_sample_slots(S), L.mn_n2v_position_slots(S)
        dev = torch.device("cuda", device)
        host_staged = dist.get_backend(group) == "gloo"
        per_max = (B + world - 1) // world
        lc = torch.empty(per_max * cap, dtype=torch.int32, device=dev)
        lt = torch.empty(per_max * cap, dtype=torch.int32, device=dev)
        le = torch.empty(per_max * cap, dtype=torch.float32, device=dev)
        lpc = torch.empty(per_max * pcap, dtype=torch.int32, device=dev)
        lpn = torch.empty(per_max * pcap * dim, dtype=torch.float32, device=dev)

        def gather(x, cnt):
            if host_staged:
                src = x[:cnt].cpu().contiguous()
                out = torch.empty(world * cnt, dtype=x.dtype)
                dist.all_gather(list(out.chunk(world)), src, group=group)
                return out.to(dev)
            out = torch.empty(world * cnt, dtype=x.dtype, device=dev)
            dist.all_gather_into_tensor(out, x[:cnt].contiguous(), group=group)
            return out

        for epoch in range(epochs):
            for w in range(num_walks):
                for b0 in range(0, n, B):
                    b1 = min(n, b0 + B)
                    per = (b1 - b0 + world - 1) // world
                    lo = min(b1, b0 + rank * per)
                    hi = min(b1, lo + per)
                    lc[:per * cap].fill_(-1)
                    lt[:per * cap].fill_(-1)
                    le[:per * cap].zero_()
                    lpc[:per * pcap].fill_(-1)
                    torch.cuda.synchronize(dev)
                    if hi > lo and L.mn_n2v_samples(S, epoch, w, lo, hi, lc.data_ptr(), lt.data_ptr(), le.data_ptr(),
                                                    lpc.data_ptr(), lpn.data_ptr()) != 0:
                        raise MuninnHipError((L.mn_node2vec_last_error() or b"").decode())
                    L.mn_n2v_sync(S)
                    gc, gt, ge = gather(lc, per * cap), gather(lt, per * cap), gather(le, per * cap)
                    gpc, gpn = gather(lpc, per * pcap), gather(lpn, per * pcap * dim)
                    torch.cuda.synchronize(dev)
                    if L.mn_n2v_apply(S, gc.data_ptr(), gt.data_ptr(), ge.data_ptr(), world * per * cap,
                                      gpc.data_ptr(), gpn.data_ptr(), world * per * pcap) != 0:
                        raise MuninnHipError((L.mn_node2vec_last_error() or b"").decode())
                    L.mn_n2v_sync(S)
        out = np.zeros((n, dim), np.float32)
        st = N2vStats()
        if L.mn_n2v_finish(S, out, C.byref(st)) < 0:
            raise MuninnHipError((L.mn_node2vec_last_error() or b"").decode())
        pairs = torch.tensor([st.pairs], dtype=torch.int64, device="cpu" if host_staged else dev)
        dist.all_reduce(pairs, group=group)
        return out, {"pairs": int(pairs.item()), "device_ms": st.device_ms, "batch_walks": B}
    finally:
        L.mn_n2v_end(S)


def build_distributed(g, ids, vectors, grow_div=16, max_batch=8192, group=None, min_split=256):
    """mn_hnsw_build on N GPUs that each hold a replica of ONE index: the batches are the same as on one GPU
    (batch <= max(1, count/grow_div), capped at max_batch); inside a batch every rank searches a contiguous slice of the
    batch's nodes against its replica (the replicas are identical, so the selected-neighbour lists are the ones a
    single GPU computes), the lists are all-gathered (RCCL over xGMI under backend "nccl": m * nlev * 2M int32, a few
    MB per batch) and every rank applies the whole batch's links.  All replicas end up with the graph of a one-GPU
    mn_hnsw_build, bit for bit.  Batches below `min_split` nodes are searched whole by every rank (no exchange).
    g: HnswIndex on this rank's device; ids / vectors: the SAME arrays on every rank."""
    import ctypes as C

    import numpy as np

    from .hnsw import MuninnHipError, _err

    L = g.L
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ids = np.ascontiguousarray(ids, np.int64)
    vectors = np.ascontiguousarray(vectors, np.float32)
    n = len(ids)
    dev = torch.device("cuda", g.device)
    host_staged = dist.get_backend(group) == "gloo"
    pos = 0
    while pos < n:
        b = max(1, g.node_count // grow_div)
        b = min(b, max_batch, n - pos)
        m = L.mn_hnsw_batch_stage(g.h, ids[pos:pos + b], vectors[pos:pos + b], b)
        if m < 0:
            raise MuninnHipError(_err())
        pos += b
        if m == 0:
            continue
        nlev, w0 = C.c_int(0), C.c_int(0)
        L.mn_hnsw_batch_dims(g.h, C.byref(nlev), C.byref(w0))
        nlev, w0 = nlev.value, w0.value
        split = world > 1 and m >= min_split
        per = (m + world - 1) // world if split else m
        rows = per * world if split else m
        sel = torch.full((rows, nlev, w0), -1, dtype=torch.int32, device=dev)
        nsel = torch.zeros((rows, nlev), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        lo, hi = (min(m, rank * per), min(m, rank * per + per)) if split else (0, m)
        if L.mn_hnsw_batch_search(g.h, lo, hi, sel.data_ptr(), nsel.data_ptr()) != 0:
            raise MuninnHipError(_err())
        if split:
            for t in (sel, nsel):
                mine = t[rank * per:(rank + 1) * per]
                if host_staged:
                    out = torch.empty((world,) + tuple(mine.shape), dtype=t.dtype)
                    dist.all_gather(list(out.unbind(0)), mine.cpu().contiguous(), group=group)
                    t.copy_(out.reshape(t.shape).to(dev))
                else:
                    dist.all_gather_into_tensor(t, mine.clone(), group=group)
            torch.cuda.synchronize(dev)
        if L.mn_hnsw_batch_link(g.h, sel.data_ptr(), nsel.data_ptr()) != 0:
            raise MuninnHipError(_err())
    return 0
