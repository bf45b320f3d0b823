"""ctypes bindings for the CPU oracle (oracle/libmn_oracle.so) and, when it has been built in this
container, for the compiled reference (oracle/_ref/libmuninn_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libmn_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmuninn_ref.so")
REF_EXT = os.path.join(HERE, "_ref", "muninn")  # sqlite3 load_extension path (no suffix)

METRIC = {"l2": 0, "cosine": 1, "inner_product": 2}
ORDER_SSE, ORDER_WAVE = 0, 1
VISITED_BITMAP, VISITED_LINEAR = 0, 1

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(target: str = "oracle") -> None:
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


class _Result(C.Structure):
    _fields_ = [("id", C.c_int64), ("distance", C.c_float)]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_dist", "n_expanded", "n_prune", "max_cand", "max_visited")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build("oracle")
        L = C.CDLL(ORACLE_SO)
        L.orc_vec_distance.restype = C.c_float
        L.orc_vec_distance.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_int]
        L.orc_dist_batch.argtypes = [C.c_int, C.c_int, _f32p, _f32p, C.c_int64, C.c_int, _f32p]
        L.orc_vec_parse_metric.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.orc_pq_trace.argtypes = [_i32p, _i64p, _f32p, C.c_int, _i64p, _f32p]
        L.orc_hnsw_create.restype = C.c_void_p
        L.orc_hnsw_create.argtypes = [C.c_int] * 4
        L.orc_hnsw_destroy.argtypes = [C.c_void_p]
        L.orc_hnsw_seed_rng.argtypes = [C.c_void_p, C.c_uint]
        L.orc_hnsw_set_order.argtypes = [C.c_void_p, C.c_int]
        L.orc_hnsw_set_visited.argtypes = [C.c_void_p, C.c_int]
        L.orc_hnsw_insert.argtypes = [C.c_void_p, C.c_int64, _f32p]
        L.orc_hnsw_insert_batch.argtypes = [C.c_void_p, _i64p, _f32p, C.c_int]
        L.orc_hnsw_search.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.POINTER(_Result)]
        L.orc_hnsw_delete.argtypes = [C.c_void_p, C.c_int64]
        L.orc_hnsw_random_level.argtypes = [C.c_void_p]
        L.orc_hnsw_node_count.argtypes = [C.c_void_p]
        L.orc_hnsw_entry_point.restype = C.c_int64
        L.orc_hnsw_entry_point.argtypes = [C.c_void_p]
        L.orc_hnsw_max_level.argtypes = [C.c_void_p]
        L.orc_hnsw_node_level.argtypes = [C.c_void_p, C.c_int64]
        L.orc_hnsw_node_deleted.argtypes = [C.c_void_p, C.c_int64]
        L.orc_hnsw_neighbors.argtypes = [C.c_void_p, C.c_int64, C.c_int, _i64p, C.c_int]
        L.orc_hnsw_get_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
        L.orc_hnsw_reset_stats.argtypes = [C.c_void_p]
        L.orc_hnsw_load_node.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_int, C.c_int]
        L.orc_hnsw_load_neighbors.argtypes = [C.c_void_p, C.c_int64, C.c_int, _i64p, C.c_int]
        L.orc_hnsw_set_entry.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.orc_hnsw_load_bulk.argtypes = [C.c_void_p, C.c_int, _i64p, _f32p, _i32p, _i32p]
        L.orc_hnsw_load_links_bulk.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int]
        _lib = L
    return _lib


def distance(metric: str, a, b, order: int = ORDER_SSE) -> np.float32:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return np.float32(lib().orc_vec_distance(METRIC[metric], order, a, b, a.shape[0]))


def dist_batch(metric: str, q, rows, order: int = ORDER_SSE) -> np.ndarray:
    q = np.ascontiguousarray(q, np.float32)
    rows = np.ascontiguousarray(rows, np.float32)
    out = np.empty(rows.shape[0], np.float32)
    lib().orc_dist_batch(METRIC[metric], order, q, rows, rows.shape[0], rows.shape[1], out)
    return out


def pq_trace(ops, ids, dists):
    ops = np.ascontiguousarray(ops, np.int32)
    ids = np.ascontiguousarray(ids, np.int64)
    dists = np.ascontiguousarray(dists, np.float32)
    oi = np.empty(len(ops), np.int64)
    od = np.empty(len(ops), np.float32)
    n = lib().orc_pq_trace(ops, ids, dists, len(ops), oi, od)
    return oi[:n].copy(), od[:n].copy()


def graph_digest(g):
    """sha256 over a graph() result: levels, then every (id, level) list in key order, entry point, top level"""
    import hashlib

    flat = list(g["levels"])
    for (i, l) in sorted(g["nbrs"]):
        flat += [i, l, len(g["nbrs"][(i, l)])] + list(g["nbrs"][(i, l)])
    flat += [g["entry"], g["max_level"]]
    return np.frombuffer(hashlib.sha256(np.array(flat, np.int64).tobytes()).digest(), np.uint8)


class _IndexBase:
    """Common Python surface over the oracle and the compiled reference."""

    def graph(self, ids):
        """{'levels': [...], 'nbrs': {(id, level): [ids...]}, 'entry':…, 'max_level':…}"""
        levels = []
        nbrs = {}
        buf = np.empty(4096, np.int64)
        for i in ids:
            lv = self.node_level(int(i))
            levels.append(lv)
            for l in range(lv + 1):
                n = self._neighbors(int(i), l, buf)
                nbrs[(int(i), l)] = buf[:n].tolist()
        return {"levels": levels, "nbrs": nbrs, "entry": self.entry_point, "max_level": self.max_level}


class Oracle(_IndexBase):
    def __init__(self, dim, metric="cosine", M=16, ef_construction=200, order=ORDER_SSE, visited=VISITED_BITMAP,
                 seed=None):
        self.L = lib()
        self.dim = dim
        self.h = self.L.orc_hnsw_create(dim, METRIC[metric], M, ef_construction)
        if order != ORDER_SSE:
            self.L.orc_hnsw_set_order(self.h, order)
        if visited != VISITED_BITMAP:
            self.L.orc_hnsw_set_visited(self.h, visited)
        if seed is not None:
            self.L.orc_hnsw_seed_rng(self.h, seed)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_hnsw_destroy(self.h)
            self.h = None

    def insert(self, id, vec):
        return self.L.orc_hnsw_insert(self.h, int(id), np.ascontiguousarray(vec, np.float32))

    def insert_many(self, ids, vecs):
        vecs = np.ascontiguousarray(vecs, np.float32)
        for i, v in zip(ids, vecs):
            if self.L.orc_hnsw_insert(self.h, int(i), v) != 0:
                return -1
        return 0

    def insert_batch(self, ids, vecs):
        ids = np.ascontiguousarray(ids, np.int64)
        vecs = np.ascontiguousarray(vecs, np.float32)
        return self.L.orc_hnsw_insert_batch(self.h, ids, vecs, len(ids))

    def search(self, q, k, ef):
        r = (_Result * k)()
        n = self.L.orc_hnsw_search(self.h, np.ascontiguousarray(q, np.float32), k, ef, r)
        return (np.array([r[i].id for i in range(n)], np.int64), np.array([r[i].distance for i in range(n)], np.float32))

    def search_many(self, Q, k, ef):
        Q = np.ascontiguousarray(Q, np.float32)
        ids = np.full((len(Q), k), -1, np.int64)
        ds = np.zeros((len(Q), k), np.float32)
        cnt = np.zeros(len(Q), np.int32)
        r = (_Result * k)()
        for qi in range(len(Q)):
            n = self.L.orc_hnsw_search(self.h, Q[qi], k, ef, r)
            cnt[qi] = n
            for i in range(n):
                ids[qi, i] = r[i].id
                ds[qi, i] = r[i].distance
        return ids, ds, cnt

    def delete(self, id):
        return self.L.orc_hnsw_delete(self.h, int(id))

    def random_level(self):
        return self.L.orc_hnsw_random_level(self.h)

    def load_node(self, id, vec, level, deleted=0):
        return self.L.orc_hnsw_load_node(self.h, int(id), np.ascontiguousarray(vec, np.float32), level, deleted)

    def load_neighbors(self, id, level, nbrs):
        nbrs = np.ascontiguousarray(nbrs, np.int64)
        return self.L.orc_hnsw_load_neighbors(self.h, int(id), level, nbrs, len(nbrs))

    def set_entry(self, entry, max_level):
        self.L.orc_hnsw_set_entry(self.h, int(entry), int(max_level))

    def load_from_device(self, g, vectors=None):
        """Mirror a device index (sqlite_muninn_amd.HnswIndex) through its bulk export."""
        ids, lv, dl = g.export_nodes()
        if vectors is None:
            vectors = g.export_vectors()
        vectors = np.ascontiguousarray(vectors, np.float32)
        assert self.L.orc_hnsw_load_bulk(self.h, len(ids), ids, vectors, lv, dl) == 0
        for l in range(int(lv.max()) + 1 if len(lv) else 0):
            rows = g.export_links(l)
            self.L.orc_hnsw_load_links_bulk(self.h, l, rows, rows.shape[1])
        self.set_entry(g.entry_point, g.max_level)

    @property
    def node_count(self):
        return self.L.orc_hnsw_node_count(self.h)

    @property
    def entry_point(self):
        return self.L.orc_hnsw_entry_point(self.h)

    @property
    def max_level(self):
        return self.L.orc_hnsw_max_level(self.h)

    def node_level(self, id):
        return self.L.orc_hnsw_node_level(self.h, id)

    def node_deleted(self, id):
        return self.L.orc_hnsw_node_deleted(self.h, id)

    def _neighbors(self, id, level, buf):
        return self.L.orc_hnsw_neighbors(self.h, id, level, buf, len(buf))

    def stats(self):
        s = _Stats()
        self.L.orc_hnsw_get_stats(self.h, C.byref(s))
        return {n: getattr(s, n) for n, _ in _Stats._fields_}

    def reset_stats(self):
        self.L.orc_hnsw_reset_stats(self.h)


# ───────────────────────── compiled reference (only in the build container) ─────────────────────────

_ref = None


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.hnsw_create.restype = C.c_void_p
        R.hnsw_create.argtypes = [C.c_int] * 4
        R.hnsw_destroy.argtypes = [C.c_void_p]
        R.hnsw_seed_rng.argtypes = [C.c_void_p, C.c_uint]
        R.hnsw_insert.argtypes = [C.c_void_p, C.c_int64, _f32p]
        R.hnsw_search.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.POINTER(_Result)]
        R.hnsw_delete.argtypes = [C.c_void_p, C.c_int64]
        R.ref_node_count.argtypes = [C.c_void_p]
        R.ref_entry_point.restype = C.c_int64
        R.ref_entry_point.argtypes = [C.c_void_p]
        R.ref_max_level.argtypes = [C.c_void_p]
        R.ref_node_level.argtypes = [C.c_void_p, C.c_int64]
        R.ref_node_deleted.argtypes = [C.c_void_p, C.c_int64]
        R.ref_neighbors.argtypes = [C.c_void_p, C.c_int64, C.c_int, _i64p, C.c_int]
        R.ref_distance.restype = C.c_float
        R.ref_distance.argtypes = [C.c_int, _f32p, _f32p, C.c_int]
        R.ref_dist_batch.argtypes = [C.c_int, _f32p, _f32p, C.c_int64, C.c_int, _f32p]
        R.ref_insert_many.argtypes = [C.c_void_p, _i64p, _f32p, C.c_int]
        R.ref_search_many.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_int, _i64p, _f32p, _i32p]
        R.ref_load_nodes.argtypes = [C.c_void_p, C.c_int64, _i64p, _f32p, _i32p, _u8p, C.c_int64, C.c_int]
        R.ref_load_edges.argtypes = [C.c_void_p, C.c_int64, _i64p, _i64p, _i32p]
        R.vec_parse_metric.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        R.pq_init.argtypes = [C.c_void_p, C.c_int]
        R.pq_push.argtypes = [C.c_void_p, C.c_int64, C.c_float]
        R.pq_pop.restype = _Result
        R.pq_pop.argtypes = [C.c_void_p]
        R.pq_destroy.argtypes = [C.c_void_p]
        _ref = R
    return _ref


class Ref(_IndexBase):
    """The reference's own hnsw_algo.c, compiled from /root/reference/src (oracle/Makefile: ref)."""

    def __init__(self, dim, metric="cosine", M=16, ef_construction=200, seed=None):
        self.R = ref_lib()
        self.dim = dim
        self.h = self.R.hnsw_create(dim, METRIC[metric], M, ef_construction)
        if seed is not None:
            self.R.hnsw_seed_rng(self.h, seed)

    def __del__(self):
        if getattr(self, "h", None):
            self.R.hnsw_destroy(self.h)
            self.h = None

    def insert(self, id, vec):
        return self.R.hnsw_insert(self.h, int(id), np.ascontiguousarray(vec, np.float32))

    def load_from_device(self, g, vectors=None):
        """Load a device-built graph into the reference's own structures through the reference's own load
        API (ref_access.c: ref_load_nodes/ref_load_edges), so its hnsw_search can be timed on that graph."""
        ids, lv, dl = g.export_nodes()
        if vectors is None:
            vectors = g.export_vectors()
        vectors = np.ascontiguousarray(vectors, np.float32)
        lv = np.ascontiguousarray(lv, np.int32)
        dl = np.ascontiguousarray(dl, np.uint8)
        assert self.R.ref_load_nodes(self.h, len(ids), ids, vectors, lv, dl, int(g.entry_point), int(g.max_level)) == 0
        for l in range(int(lv.max()) + 1 if len(lv) else 0):
            rows = g.export_links(l)
            r, c = np.nonzero(rows >= 0)  # row-major: keeps each list's order
            src = np.ascontiguousarray(ids[r])
            dst = np.ascontiguousarray(ids[rows[r, c]])
            lev = np.full(len(src), l, np.int32)
            self.R.ref_load_edges(self.h, len(src), src, dst, lev)

    def insert_many(self, ids, vecs):
        ids = np.ascontiguousarray(ids, np.int64)
        vecs = np.ascontiguousarray(vecs, np.float32)
        return 0 if self.R.ref_insert_many(self.h, ids, vecs, len(ids)) == len(ids) else -1

    def search(self, q, k, ef):
        r = (_Result * k)()
        n = self.R.hnsw_search(self.h, np.ascontiguousarray(q, np.float32), k, ef, r)
        return (np.array([r[i].id for i in range(n)], np.int64), np.array([r[i].distance for i in range(n)], np.float32))

    def search_many(self, Q, k, ef):
        Q = np.ascontiguousarray(Q, np.float32)
        ids = np.empty((len(Q), k), np.int64)
        ds = np.empty((len(Q), k), np.float32)
        cnt = np.empty(len(Q), np.int32)
        self.R.ref_search_many(self.h, Q, len(Q), k, ef, ids, ds, cnt)
        return ids, ds, cnt

    def delete(self, id):
        return self.R.hnsw_delete(self.h, int(id))

    def add_neighbors(self, id, level, nbrs):
        """node_add_neighbor(id, level, nbr) for every nbr, as the shadow-table loader does (src/hnsw_vtab.c:333-336)."""
        nbrs = np.ascontiguousarray(nbrs, np.int64)
        src = np.full(len(nbrs), int(id), np.int64)
        lev = np.full(len(nbrs), int(level), np.int32)
        return self.R.ref_load_edges(self.h, len(nbrs), src, nbrs, lev)

    @property
    def node_count(self):
        return self.R.ref_node_count(self.h)

    @property
    def entry_point(self):
        return self.R.ref_entry_point(self.h)

    @property
    def max_level(self):
        return self.R.ref_max_level(self.h)

    def node_level(self, id):
        return self.R.ref_node_level(self.h, id)

    def node_deleted(self, id):
        return self.R.ref_node_deleted(self.h, id)

    def _neighbors(self, id, level, buf):
        return self.R.ref_neighbors(self.h, id, level, buf, len(buf))


def ref_distance(metric, a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return np.float32(ref_lib().ref_distance(METRIC[metric], a, b, a.shape[0]))


def ref_dist_batch(metric, q, rows):
    q = np.ascontiguousarray(q, np.float32)
    rows = np.ascontiguousarray(rows, np.float32)
    out = np.empty(rows.shape[0], np.float32)
    ref_lib().ref_dist_batch(METRIC[metric], q, rows, rows.shape[0], rows.shape[1], out)
    return out


def ref_pq_trace(ops, ids, dists):
    """Replay a push/pop trace through the reference's priority_queue.c."""
    R = ref_lib()
    pq = C.create_string_buffer(32)  # PriorityQueue {ptr, int, int} = 16 bytes; over-allocated
    R.pq_init(pq, 4)
    oi, od = [], []
    size = 0
    for op, i, d in zip(ops, ids, dists):
        if op:
            R.pq_push(pq, int(i), float(d))
            size += 1
        elif size > 0:
            it = R.pq_pop(pq)
            oi.append(it.id)
            od.append(it.distance)
            size -= 1
    R.pq_destroy(pq)
    return np.array(oi, np.int64), np.array(od, np.float32)
