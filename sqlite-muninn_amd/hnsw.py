"""Host-side mirror of the reference's hnsw_algo.h / vec_math.h over libmuninn_hip.so (ctypes).

Same names, argument meaning and error behaviour as the reference (src/hnsw_algo.h:55-92):
insert/delete return 0 / -1, search returns the hits found, create raises if no device.
Nothing here computes: every call lands in the HIP library; if the library or a gfx950 device
is missing the call fails loudly (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .build import LIB

METRIC = {"l2": 0, "cosine": 1, "inner_product": 2}  # src/vec_math.h:13
ORDER_SSE, ORDER_WAVE = 0, 1
BUILD_SEQUENTIAL, BUILD_BATCHED = 0, 1

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class MuninnHipError(RuntimeError):
    pass


class _Result(C.Structure):
    _fields_ = [("id", C.c_int64), ("distance", C.c_float)]


class LaunchStats(C.Structure):
    _fields_ = [("last_kernel_ms", C.c_double), ("last_n_dist", C.c_int64), ("last_n_expanded", C.c_int64),
                ("last_n_overflow", C.c_int64)]


class BuildStats(C.Structure):
    _fields_ = [("search_ms", C.c_double), ("link_ms", C.c_double), ("n_dist", C.c_int64), ("n_expanded", C.c_int64),
                ("batches", C.c_int64), ("nodes", C.c_int64)]


# every symbol include/muninn_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mn_abi_version", C.c_int, []),
    ("mn_last_error", C.c_char_p, []),
    ("mn_device_count", C.c_int, []),
    ("mn_debug_fault_alloc", C.c_longlong, [C.c_longlong]),
    ("mn_vec_parse_metric", C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    ("mn_vec_dist_batch", C.c_int, [C.c_int, C.c_int, _f32p, _f32p, C.c_int64, C.c_int, _f32p]),
    ("mn_hnsw_create", C.c_void_p, [C.c_int] * 4),
    ("mn_hnsw_create_on", C.c_void_p, [C.c_int] * 5),
    ("mn_hnsw_destroy", None, [C.c_void_p]),
    ("mn_hnsw_seed_rng", None, [C.c_void_p, C.c_uint]),
    ("mn_hnsw_set_order", C.c_int, [C.c_void_p, C.c_int]),
    ("mn_hnsw_insert", C.c_int, [C.c_void_p, C.c_int64, _f32p]),
    ("mn_hnsw_insert_batch", C.c_int, [C.c_void_p, _i64p, _f32p, C.c_int64, C.c_int]),
    ("mn_hnsw_insert_logged", C.c_int, [C.c_void_p, C.c_int64, _f32p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("mn_hnsw_log_invalidate", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("mn_hnsw_build", C.c_int, [C.c_void_p, _i64p, _f32p, C.c_int64, C.c_int, C.c_int]),
    ("mn_hnsw_build_dev", C.c_int, [C.c_void_p, _i64p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    ("mn_hnsw_device", C.c_int, [C.c_void_p]),
    ("mn_hnsw_search", C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.POINTER(_Result)]),
    ("mn_hnsw_search_batch", C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, C.c_int, _i64p, _f32p, _i32p]),
    ("mn_hnsw_search_batch_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mn_hnsw_sync", C.c_int, [C.c_void_p]),
    ("mn_hnsw_delete", C.c_int, [C.c_void_p, C.c_int64]),
    ("mn_hnsw_get_vector", C.c_int, [C.c_void_p, C.c_int64, _f32p]),
    ("mn_hnsw_node_count", C.c_int, [C.c_void_p]),
    ("mn_hnsw_entry_point", C.c_int64, [C.c_void_p]),
    ("mn_hnsw_max_level", C.c_int, [C.c_void_p]),
    ("mn_hnsw_node_level", C.c_int, [C.c_void_p, C.c_int64]),
    ("mn_hnsw_node_deleted", C.c_int, [C.c_void_p, C.c_int64]),
    ("mn_hnsw_neighbors", C.c_int, [C.c_void_p, C.c_int64, C.c_int, _i64p, C.c_int]),
    ("mn_hnsw_load_node", C.c_int, [C.c_void_p, C.c_int64, _f32p, C.c_int, C.c_int]),
    ("mn_hnsw_load_neighbors", C.c_int, [C.c_void_p, C.c_int64, C.c_int, _i64p, C.c_int]),
    ("mn_hnsw_set_entry", C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    ("mn_hnsw_slot_count", C.c_int, [C.c_void_p]),
    ("mn_hnsw_export_nodes", C.c_int, [C.c_void_p, _i64p, _i32p, _i32p]),
    ("mn_hnsw_export_vectors", C.c_int, [C.c_void_p, _f32p]),
    ("mn_hnsw_export_links", C.c_int, [C.c_void_p, C.c_int, _i32p, C.POINTER(C.c_int)]),
    ("mn_hnsw_row_width", C.c_int, [C.c_void_p, C.c_int]),
    ("mn_hnsw_batch_stage", C.c_int, [C.c_void_p, _i64p, _f32p, C.c_int64]),
    ("mn_hnsw_batch_dims", C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("mn_hnsw_batch_search", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("mn_hnsw_batch_link", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mn_hnsw_take_dirty", C.c_int64, [C.c_void_p, _i64p, C.c_int64]),
    ("mn_hnsw_edges_of", C.c_int64, [C.c_void_p, _i64p, C.c_int, _i64p, _i64p, _i32p, _f32p, C.c_int64]),
    ("mn_hnsw_last_launch", C.c_int, [C.c_void_p, C.POINTER(LaunchStats)]),
    ("mn_hnsw_build_stats", C.c_int, [C.c_void_p, C.POINTER(BuildStats), C.c_int]),
    ("mn_dev_malloc", C.c_void_p, [C.c_void_p, C.c_size_t]),
    ("mn_dev_free", None, [C.c_void_p, C.c_void_p]),
    ("mn_dev_upload", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("mn_dev_download", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("mn_hnsw_bruteforce_topk", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, _i64p]),
    # multi-GPU
    ("mn_comm_unique_id", C.c_int, [C.c_void_p]),
    ("mn_comm_init_rccl", C.c_void_p, [C.c_int, C.c_int, C.c_void_p, C.c_int]),
    ("mn_comm_init_host", C.c_void_p, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    ("mn_comm_world", C.c_int, [C.c_void_p]),
    ("mn_comm_rank", C.c_int, [C.c_void_p]),
    ("mn_comm_destroy", None, [C.c_void_p]),
    ("mn_comm_last_error", C.c_char_p, []),
    ("mn_hnsw_build_shared", C.c_int, [C.c_void_p, C.c_void_p, _i64p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_int]),
    ("mn_hnsw_search_sharded_dev", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    ("mn_hnsw_search_sharded", C.c_int, [C.c_void_p, C.c_void_p, _f32p, C.c_int64, C.c_int, C.c_int, _i64p, _f32p, _i32p]),
    # the sharded index inside one process
    ("mn_shards_create", C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]),
    ("mn_shards_destroy", None, [C.c_void_p]),
    ("mn_shards_count", C.c_int, [C.c_void_p]),
    ("mn_shards_index", C.c_void_p, [C.c_void_p, C.c_int]),
    ("mn_shards_of", C.c_int, [C.c_void_p, C.c_int64]),
    ("mn_shards_set_order", C.c_int, [C.c_void_p, C.c_int]),
    ("mn_shards_insert", C.c_int, [C.c_void_p, C.c_int64, _f32p]),
    ("mn_shards_delete", C.c_int, [C.c_void_p, C.c_int64]),
    ("mn_shards_build", C.c_int, [C.c_void_p, _i64p, _f32p, C.c_int64, C.c_int, C.c_int]),
    ("mn_shards_search", C.c_int, [C.c_void_p, _f32p, C.c_int, C.c_int, C.c_void_p]),
    ("mn_shards_search_batch", C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_int, C.c_int, _i64p, _f32p, _i32p]),
    ("mn_shards_last_error", C.c_char_p, []),
]

_lib = None


def lib():
    """Loads libmuninn_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise MuninnHipError(f"{LIB} is missing: run __graft_entry__.build() (hipcc, gfx950)")
        L = C.CDLL(LIB)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _err() -> str:
    m = lib().mn_last_error()
    return m.decode() if m else ""


def device_count() -> int:
    return lib().mn_device_count()


def vec_parse_metric(name: str) -> int:
    """vec_parse_metric (src/vec_math.c:192-204): metric enum, or -1."""
    out = C.c_int(-1)
    rc = lib().mn_vec_parse_metric(name.encode(), C.byref(out))
    return out.value if rc == 0 else -1


def vec_dist_batch(metric: str, query, rows, order: int = ORDER_SSE) -> np.ndarray:
    """vec_get_distance_func(metric)(query, rows[i], dim) for every row, on the device."""
    q = np.ascontiguousarray(query, np.float32)
    r = np.ascontiguousarray(rows, np.float32).reshape(-1, q.shape[0])
    out = np.empty(r.shape[0], np.float32)
    if lib().mn_vec_dist_batch(METRIC[metric], order, q, r, r.shape[0], q.shape[0], out) != 0:
        raise MuninnHipError(_err())
    return out


class ShardedIndex:
    """BASELINE config 3 inside one process (mn_shards_*): rowid mod n -> one HNSW graph per entry of `devices`."""

    def __init__(self, dim, metric="cosine", M=16, ef_construction=200, devices=(0,), order=ORDER_SSE):
        self.L = lib()
        self.dim = dim
        dv = np.ascontiguousarray(devices, np.int32)
        self.h = self.L.mn_shards_create(dim, METRIC[metric] if isinstance(metric, str) else metric, M, ef_construction, dv,
                                         len(dv))
        if not self.h:
            raise MuninnHipError("mn_shards_create failed: " + self._err())
        if order != ORDER_SSE and self.L.mn_shards_set_order(self.h, order) != 0:
            raise MuninnHipError(self._err())

    def _err(self):
        return (self.L.mn_shards_last_error() or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self.L.mn_shards_destroy(self.h)
            self.h = None

    __del__ = close

    def insert(self, id, vec) -> int:
        return self.L.mn_shards_insert(self.h, int(id), np.ascontiguousarray(vec, np.float32))

    def delete(self, id) -> int:
        return self.L.mn_shards_delete(self.h, int(id))

    def build(self, ids, vectors, grow_div=16, max_batch=8192) -> int:
        ids = np.ascontiguousarray(ids, np.int64)
        vectors = np.ascontiguousarray(vectors, np.float32).reshape(len(ids), self.dim)
        return self.L.mn_shards_build(self.h, ids, vectors, len(ids), grow_div, max_batch)

    def search(self, q, k, ef):
        r = (_Result * max(k, 1))()
        n = self.L.mn_shards_search(self.h, np.ascontiguousarray(q, np.float32), k, ef, r)
        if n < 0:
            raise MuninnHipError(self._err())
        return (np.array([r[i].id for i in range(n)], np.int64), np.array([r[i].distance for i in range(n)], np.float32))

    def search_batch(self, Q, k, ef):
        Q = np.ascontiguousarray(Q, np.float32).reshape(-1, self.dim)
        nq = len(Q)
        ids = np.full((nq, k), -1, np.int64)
        ds = np.zeros((nq, k), np.float32)
        cnt = np.zeros(nq, np.int32)
        if self.L.mn_shards_search_batch(self.h, Q, nq, k, ef, ids, ds, cnt) != 0:
            raise MuninnHipError(self._err())
        return ids, ds, cnt


class _EdgeChange(C.Structure):  # mn_edge_change
    _fields_ = [("op", C.c_int), ("level", C.c_int), ("src", C.c_int64), ("dst", C.c_int64), ("distance", C.c_float)]


class HnswIndex:
    """Device-resident HNSW index; mirrors HnswIndex + hnsw_* of src/hnsw_algo.h."""

    def __init__(self, dim, metric="cosine", M=16, ef_construction=200, order=ORDER_SSE, seed=None, device=0):
        self.L = lib()
        self.dim = dim
        self.M = M
        self.device = device
        self.h = self.L.mn_hnsw_create_on(dim, METRIC[metric] if isinstance(metric, str) else metric, M, ef_construction,
                                          device)
        if not self.h:
            raise MuninnHipError("hnsw_create failed: " + _err())
        if order != ORDER_SSE and self.L.mn_hnsw_set_order(self.h, order) != 0:
            raise MuninnHipError(_err())
        if seed is not None:
            self.L.mn_hnsw_seed_rng(self.h, seed)

    def close(self):
        if getattr(self, "h", None):
            self.L.mn_hnsw_destroy(self.h)
            self.h = None

    __del__ = close

    # ---- hnsw_algo.h surface ----
    def seed_rng(self, seed):
        self.L.mn_hnsw_seed_rng(self.h, int(seed))

    def insert(self, id, vec) -> int:
        return self.L.mn_hnsw_insert(self.h, int(id), np.ascontiguousarray(vec, np.float32))

    def insert_logged(self, id, vec, cap=1024):
        """mn_hnsw_insert_logged → (rc, changes): changes = list of (op, level, src, dst, distance) — op 1 edge added, 2 removed
        — or None when the log could not describe the insert (the persist set is then complete instead)."""
        log = (_EdgeChange * cap)()
        n = C.c_int(-1)
        rc = self.L.mn_hnsw_insert_logged(self.h, int(id), np.ascontiguousarray(vec, np.float32), log, cap, C.byref(n))
        if rc != 0 or n.value < 0:
            return rc, None
        return rc, [(log[i].op, log[i].level, log[i].src, log[i].dst, log[i].distance) for i in range(n.value)]

    def log_invalidate(self, ids=None) -> int:
        if ids is None:
            return self.L.mn_hnsw_log_invalidate(self.h, None, 0)
        ids = np.ascontiguousarray(ids, np.int64)
        return self.L.mn_hnsw_log_invalidate(self.h, ids.ctypes.data, len(ids))

    def insert_batch(self, ids, vecs, mode=BUILD_BATCHED) -> int:
        ids = np.ascontiguousarray(ids, np.int64)
        vecs = np.ascontiguousarray(vecs, np.float32)
        return self.L.mn_hnsw_insert_batch(self.h, ids, vecs, len(ids), mode)

    def build(self, ids, vecs, grow_div=16, max_batch=8192) -> int:
        ids = np.ascontiguousarray(ids, np.int64)
        vecs = np.ascontiguousarray(vecs, np.float32)
        return self.L.mn_hnsw_build(self.h, ids, vecs, len(ids), grow_div, max_batch)

    def search(self, q, k, ef):
        r = (_Result * max(k, 1))()
        n = self.L.mn_hnsw_search(self.h, np.ascontiguousarray(q, np.float32), k, ef, r)
        return (np.array([r[i].id for i in range(n)], np.int64), np.array([r[i].distance for i in range(n)], np.float32))

    def search_batch(self, Q, k, ef):
        Q = np.ascontiguousarray(Q, np.float32).reshape(-1, self.dim)
        ids = np.empty((len(Q), k), np.int64)
        ds = np.empty((len(Q), k), np.float32)
        cnt = np.empty(len(Q), np.int32)
        if self.L.mn_hnsw_search_batch(self.h, Q, len(Q), k, ef, ids, ds, cnt) != 0:
            raise MuninnHipError(_err())
        return ids, ds, cnt

    search_many = search_batch

    def delete(self, id) -> int:
        return self.L.mn_hnsw_delete(self.h, int(id))

    def get_vector(self, id):
        out = np.empty(self.dim, np.float32)
        return out if self.L.mn_hnsw_get_vector(self.h, int(id), out) == 0 else None

    # ---- state ----
    @property
    def node_count(self):
        return self.L.mn_hnsw_node_count(self.h)

    @property
    def entry_point(self):
        return self.L.mn_hnsw_entry_point(self.h)

    @property
    def max_level(self):
        return self.L.mn_hnsw_max_level(self.h)

    def node_level(self, id):
        return self.L.mn_hnsw_node_level(self.h, int(id))

    def node_deleted(self, id):
        return self.L.mn_hnsw_node_deleted(self.h, int(id))

    def neighbors(self, id, level):
        buf = np.empty(max(128, self.L.mn_hnsw_row_width(self.h, level)), np.int64)
        n = self.L.mn_hnsw_neighbors(self.h, int(id), level, buf, len(buf))
        return None if n < 0 else buf[:n].tolist()

    def graph(self, ids):
        levels, nbrs = [], {}
        for i in ids:
            lv = self.node_level(int(i))
            levels.append(lv)
            for l in range(lv + 1):
                nbrs[(int(i), l)] = self.neighbors(int(i), l)
        return {"levels": levels, "nbrs": nbrs, "entry": self.entry_point, "max_level": self.max_level}

    def load_node(self, id, vec, level, deleted=0):
        return self.L.mn_hnsw_load_node(self.h, int(id), np.ascontiguousarray(vec, np.float32), level, deleted)

    def load_neighbors(self, id, level, nbrs):
        nbrs = np.ascontiguousarray(nbrs, np.int64)
        return self.L.mn_hnsw_load_neighbors(self.h, int(id), level, nbrs, len(nbrs))

    def set_entry(self, entry, max_level):
        return self.L.mn_hnsw_set_entry(self.h, int(entry), int(max_level))

    def load_graph_from(self, other, ids, vectors):
        """Copies nodes, neighbour lists and entry point out of any object with the oracle's
        inspection surface (oracle.orc.Oracle / Ref) — used by parity tests only."""
        for i, v in zip(ids, vectors):
            if self.load_node(int(i), v, other.node_level(int(i)), max(0, other.node_deleted(int(i)))) != 0:
                raise MuninnHipError(_err())
        buf = np.empty(4096, np.int64)
        for i in ids:
            for l in range(other.node_level(int(i)) + 1):
                n = other._neighbors(int(i), l, buf)
                if self.load_neighbors(int(i), l, buf[:n].copy()) != 0:
                    raise MuninnHipError(_err())
        self.set_entry(other.entry_point, other.max_level)

    # ---- bulk export ----
    @property
    def slot_count(self):
        return self.L.mn_hnsw_slot_count(self.h)

    def export_nodes(self):
        n = self.slot_count
        ids = np.empty(n, np.int64)
        lv = np.empty(n, np.int32)
        dl = np.empty(n, np.int32)
        if self.L.mn_hnsw_export_nodes(self.h, ids, lv, dl) != 0:
            raise MuninnHipError(_err())
        return ids, lv, dl

    def export_vectors(self):
        out = np.empty((self.slot_count, self.dim), np.float32)
        if self.L.mn_hnsw_export_vectors(self.h, out) != 0:
            raise MuninnHipError(_err())
        return out

    def export_links(self, level):
        n = self.slot_count
        W = self.L.mn_hnsw_row_width(self.h, level)  # 2M / M, or more once a delete or a loaded database grew a list
        out = np.empty((n, W), np.int32)
        w = C.c_int(0)
        if self.L.mn_hnsw_export_links(self.h, level, out, C.byref(w)) != 0:
            raise MuninnHipError(_err())
        assert w.value == W
        return out

    def take_dirty(self):
        """ids the reference's xUpdate would have re-persisted since the last call (src/hnsw_vtab.c:755-768)."""
        cap = 1024
        while True:
            out = np.empty(cap, np.int64)
            n = self.L.mn_hnsw_take_dirty(self.h, out, cap)
            if n < 0:
                raise MuninnHipError(_err())
            if n <= cap:
                return out[:n]
            cap = int(n)

    def edges_of(self, ids):
        ids = np.ascontiguousarray(ids, np.int64)
        cap = max(64, len(ids) * 4 * self.M)
        while True:
            src = np.empty(cap, np.int64)
            dst = np.empty(cap, np.int64)
            lvl = np.empty(cap, np.int32)
            dist = np.empty(cap, np.float32)
            n = self.L.mn_hnsw_edges_of(self.h, ids, len(ids), src, dst, lvl, dist, cap)
            if n < 0:
                raise MuninnHipError(_err())
            if n <= cap:
                return src[:n], dst[:n], lvl[:n], dist[:n]
            cap = n

    # ---- measurement ----
    def last_launch(self):
        s = LaunchStats()
        if self.L.mn_hnsw_last_launch(self.h, C.byref(s)) != 0:
            raise MuninnHipError(_err())
        return {n: getattr(s, n) for n, _ in LaunchStats._fields_}

    def build_stats(self, reset=False):
        s = BuildStats()
        self.L.mn_hnsw_build_stats(self.h, C.byref(s), 1 if reset else 0)
        return {n: getattr(s, n) for n, _ in BuildStats._fields_}

    def dev_malloc(self, nbytes):
        p = self.L.mn_dev_malloc(self.h, nbytes)
        if not p:
            raise MuninnHipError(_err())
        return p

    def dev_free(self, p):
        self.L.mn_dev_free(self.h, p)

    def dev_upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        if self.L.mn_dev_upload(self.h, dptr, arr.ctypes.data, arr.nbytes) != 0:
            raise MuninnHipError(_err())

    def dev_download(self, arr, dptr):
        if self.L.mn_dev_download(self.h, arr.ctypes.data, dptr, arr.nbytes) != 0:
            raise MuninnHipError(_err())

    def search_batch_dev(self, d_q, nq, k, ef, d_ids, d_dists, d_counts):
        if self.L.mn_hnsw_search_batch_dev(self.h, d_q, nq, k, ef, d_ids, d_dists, d_counts) != 0:
            raise MuninnHipError(_err())

    def sync(self):
        if self.L.mn_hnsw_sync(self.h) != 0:
            raise MuninnHipError(_err())

    def bruteforce_topk(self, d_q, nq, k):
        out = np.empty((nq, k), np.int64)
        if self.L.mn_hnsw_bruteforce_topk(self.h, d_q, nq, k, out) != 0:
            raise MuninnHipError(_err())
        return out
