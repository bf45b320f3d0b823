"""Multi-GPU callers: one process per GPU.  The exchange itself lives below the C-ABI
(include/muninn_hip.h "multi-GPU": mn_comm + mn_hnsw_build_shared / mn_hnsw_search_sharded / mn_node2vec_train_shared:
RCCL all-gather over xGMI on the library's own HIP stream, merge kernels on the device).  torch.distributed is used
here for what a launcher is for — rendezvous (handing rank 0's RCCL unique id to the other ranks), barriers and the
max-over-ranks of a timing — and, under backend "gloo", as the host transport of the rehearsals in which several ranks
share one GPU (RCCL refuses that).

* kNN queries are independent → replicas + sharded query batches need NO data-path collective.
* A sharded index (config 3: rowid mod world → one HNSW graph per GPU) has exactly one exchange step:
  every rank searches the same queries on its shard, then the per-shard top-k lists — k x (int64 id,
  f32 distance) per query — are all-gathered and merged in the total order (distance, shard rank, position).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

_HOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


def shard_of_rowid(rowid: int, world: int) -> int:
    return int(rowid) % world


class Comm:
    """mn_comm for this rank of a torch.distributed group: RCCL (backend "nccl") or the host transport ("gloo")."""

    def __init__(self, device=0, group=None):
        from .hnsw import MuninnHipError, lib

        self.L = lib()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if dist.get_backend(group) == "nccl":
            box = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                # (a failure is broadcast too: the other ranks are already waiting in the rendezvous below)
                box = [buf.raw if self.L.mn_comm_unique_id(buf) == 0 else (self.L.mn_comm_last_error() or b"?").decode()]
            dist.broadcast_object_list(box, src=0, group=group)  # rendezvous only: 128 bytes
            if not isinstance(box[0], bytes):
                raise MuninnHipError(f"mn_comm_unique_id failed on rank 0: {box[0]}")
            self._id = C.create_string_buffer(box[0], 128)
            self.h = self.L.mn_comm_init_rccl(self.world, self.rank, self._id, device)
            self._cb = None
        else:
            def allgather(user, send, recv, nbytes):  # HOST buffers; recv = world x nbytes in rank order
                try:
                    src = torch.frombuffer((C.c_ubyte * nbytes).from_address(send), dtype=torch.uint8).clone()
                    out = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                    dist.all_gather(out, src, group=group)
                    joined = torch.cat(out)  # (named: it must outlive the copy)
                    C.memmove(recv, joined.data_ptr(), nbytes * self.world)
                    return 0
                except Exception:  # never let an exception cross the C boundary
                    return -1

            self._cb = _HOST_FN(allgather)
            self.h = self.L.mn_comm_init_host(self.world, self.rank, C.cast(self._cb, C.c_void_p), None, device)
        if not self.h:
            raise MuninnHipError((self.L.mn_comm_last_error() or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.mn_comm_destroy(self.h)
            self.h = None

    __del__ = close


def build_distributed(g, ids, vectors, grow_div=16, max_batch=8192, group=None, min_split=256, comm=None):
    """mn_hnsw_build on the group's GPUs, each holding a replica of ONE index (mn_hnsw_build_shared): the search half of
    every batch is split over the ranks, the selected-neighbour lists are all-gathered, every replica links the whole
    batch.  All replicas end up with the graph of a one-GPU mn_hnsw_build, bit for bit.
    g: HnswIndex on this rank's device; ids / vectors: the SAME arrays on every rank."""
    from .hnsw import MuninnHipError, _err

    own = comm is None
    if own:
        comm = Comm(g.device, group)
    try:
        ids = np.ascontiguousarray(ids, np.int64)
        vectors = np.ascontiguousarray(vectors, np.float32)
        if g.L.mn_hnsw_build_shared(g.h, comm.h, ids, vectors, len(ids), grow_div, max_batch, min_split) != 0:
            raise MuninnHipError(_err())
    finally:
        if own:
            comm.close()
    return 0


def search_sharded_dev(g, comm, d_queries, nq, k, ef, d_ids, d_dists, d_counts):
    """config 3: this rank's shard is searched, the per-shard top-k are all-gathered and merged on the device; every rank
    receives the global top-k in its (device) output buffers.  Asynchronous on the index's stream."""
    from .hnsw import MuninnHipError, _err

    if g.L.mn_hnsw_search_sharded_dev(g.h, comm.h, d_queries, nq, k, ef, d_ids, d_dists, d_counts) != 0:
        raise MuninnHipError(_err())


def search_sharded(g, comm, Q, k, ef):
    from .hnsw import MuninnHipError, _err

    Q = np.ascontiguousarray(Q, np.float32)
    ids = np.empty((len(Q), k), np.int64)
    ds = np.empty((len(Q), k), np.float32)
    cnt = np.empty(len(Q), np.int32)
    if g.L.mn_hnsw_search_sharded(g.h, comm.h, Q, len(Q), k, ef, ids, ds, cnt) != 0:
        raise MuninnHipError(_err())
    return ids, ds, cnt


def node2vec_train_distributed(off, adj, dim, p=1.0, q=1.0, num_walks=10, walk_length=80, window=5, neg_samples=5,
                               learning_rate=0.025, epochs=1, batch_walks=0, device=0, group=None, comm=None):
    """Data-parallel Node2Vec (config 4, mn_node2vec_train_shared): every rank holds a replica of syn0/syn1neg; the walks
    of each batch are split over the ranks, every (centre, target, err) sample travels to the rank that owns its target row and
    every per-position neu1e vector to the owner of its centre row (all-to-all by destination shard, buckets in walk order),
    each rank applies what it received to its rows and the updated row shards are all-gathered — the embeddings are
    bit-identical to mn_node2vec_train(..., MN_N2V_BATCHED) on one GPU.  Returns (embeddings [n][dim] float32, stats)."""
    from .graph import N2vParams, N2vStats, _glib
    from .hnsw import MuninnHipError

    L = _glib()
    own = comm is None
    if own:
        comm = Comm(device, group)
    try:
        off = np.ascontiguousarray(off, np.int32)
        adj = np.ascontiguousarray(adj if len(adj) else np.zeros(1, np.int32), np.int32)
        n = len(off) - 1
        prm = N2vParams(dim, p, q, num_walks, walk_length, window, neg_samples, learning_rate, epochs, batch_walks)
        out = np.zeros((n, dim), np.float32)
        st = N2vStats()
        if L.mn_node2vec_train_shared(comm.h, n, off, adj, C.byref(prm), device, out, C.byref(st)) < 0:
            raise MuninnHipError((L.mn_node2vec_last_error() or b"").decode())
        host = dist.get_backend(group) == "gloo"
        pairs = torch.tensor([st.pairs], dtype=torch.int64, device="cpu" if host else torch.device("cuda", device))
        dist.all_reduce(pairs, group=group)
        return out, {"pairs": int(pairs.item()), "device_ms": st.device_ms}
    finally:
        if own:
            comm.close()


def leiden_distributed(graph, resolution=1.0, direction="both", batch=0, group=None, comm=None):
    """run_leiden on the group's GPUs (mn_graph_leiden_shared): every rank holds `graph` (a graph.Graph of the SAME edges on
    its own device); the sweeps' evaluation and the modularity's per-node terms are divided by node range and all-gathered;
    every rank gets the communities and Q of a one-GPU MN_LEIDEN_BATCHED run, bit for bit.  → (community[n], Q, stats)."""
    from .graph import LeidenStats, MuninnHipError, _gerr

    own = comm is None
    if own:
        comm = Comm(0, group)
    try:
        comm_arr = np.empty(max(graph.n, 1), np.int32)
        q = C.c_double(0.0)
        rc = graph.L.mn_graph_leiden_shared(graph.h, comm.h, float(resolution), 1 if direction == "both" else 0, int(batch), comm_arr,
                                            C.byref(q))
        if rc != 0:
            raise MuninnHipError(_gerr())
        st = LeidenStats()
        graph.L.mn_graph_leiden_stats(graph.h, C.byref(st))
        return comm_arr[:graph.n], q.value, {n: getattr(st, n) for n, _ in LeidenStats._fields_}
    finally:
        if own:
            comm.close()


def allgather_merge_topk(ids: torch.Tensor, dists: torch.Tensor, counts: torch.Tensor, k: int, group=None):
    """The merge rule of the sharded index restated with torch ops — used by the CPU (gloo) rehearsal of the exchange step,
    where no HIP device exists; the product path is mn_hnsw_search_sharded (k_merge_topk on the device).
    ids [nq,k] int64 (-1 padded), dists [nq,k] f32, counts [nq] int32 — this rank's per-shard results (ascending by
    distance).  Returns the merged global top-k (ids, dists, counts) on every rank."""
    world = dist.get_world_size(group)
    nq = ids.shape[0]
    dev = ids.device
    staged = dist.get_backend(group) == "gloo" and dev.type != "cpu"
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
    order = torch.sort(key, dim=1, stable=True).indices[:, :k]  # stable: (shard rank, position) among equal distances
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
