"""Host-side mirror of the reference's graph hot path over libmuninn_hip.so: device-resident CSR
adjacency (src/graph_csr.h) and run_leiden (src/graph_community.c:336).  ctypes only; no compute here."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .hnsw import MuninnHipError, lib

LEIDEN_SEQUENTIAL, LEIDEN_BATCHED = 0, 1

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class LeidenStats(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("moves", C.c_int64), ("move_sweeps", C.c_int64), ("refine_sweeps", C.c_int64),
                ("n_communities", C.c_int), ("device_ms", C.c_double)]


class N2vParams(C.Structure):
    _fields_ = [("dim", C.c_int), ("p", C.c_double), ("q", C.c_double), ("num_walks", C.c_int), ("walk_length", C.c_int),
                ("window", C.c_int), ("neg_samples", C.c_int), ("learning_rate", C.c_double), ("epochs", C.c_int),
                ("batch_walks", C.c_int)]


class N2vStats(C.Structure):
    _fields_ = [("pairs", C.c_int64), ("device_ms", C.c_double)]


N2V_SEQUENTIAL, N2V_BATCHED = 0, 1
COMPONENTS_EXACT, COMPONENTS_FAST = 0, 1


class AlgoStats(C.Structure):
    _fields_ = [("device_ms", C.c_double), ("iterations", C.c_int), ("aux", C.c_int64)]


GRAPH_SYMBOLS = [
    ("mn_node2vec_train", C.c_int, [C.c_int, _i32p, _i32p, C.POINTER(N2vParams), C.c_int, C.c_int,
                                    np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.POINTER(N2vStats)]),
    ("mn_node2vec_last_error", C.c_char_p, []),
    ("mn_node2vec_train_shared", C.c_int, [C.c_void_p, C.c_int, _i32p, _i32p, C.POINTER(N2vParams), C.c_int,
                                           np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.POINTER(N2vStats)]),
    ("mn_n2v_begin", C.c_void_p, [C.c_int, _i32p, _i32p, C.POINTER(N2vParams), C.c_int]),
    ("mn_n2v_batch_walks", C.c_int, [C.c_void_p]),
    ("mn_n2v_sample_slots", C.c_int, [C.c_void_p]),
    ("mn_n2v_position_slots", C.c_int, [C.c_void_p]),
    ("mn_n2v_samples", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    ("mn_n2v_apply", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]),
    ("mn_n2v_sync", C.c_int, [C.c_void_p]),
    ("mn_n2v_finish", C.c_int, [C.c_void_p, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.POINTER(N2vStats)]),
    ("mn_n2v_finish_dev", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(N2vStats)]),
    ("mn_node2vec_train_into", C.c_int, [C.c_int, _i32p, _i32p, C.POINTER(N2vParams), C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.POINTER(N2vStats), C.POINTER(C.c_double)]),
    ("mn_n2v_end", None, [C.c_void_p]),
    ("mn_graph_create", C.c_void_p, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("mn_graph_create_blocked", C.c_void_p, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    ("mn_graph_destroy", None, [C.c_void_p]),
    ("mn_graph_last_error", C.c_char_p, []),
    ("mn_graph_leiden", C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, _i32p, C.POINTER(C.c_double)]),
    ("mn_graph_leiden_stats", C.c_int, [C.c_void_p, C.POINTER(LeidenStats)]),
    ("mn_graph_leiden_shared", C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, _i32p, C.POINTER(C.c_double)]),
    ("mn_graph_betweenness", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"),
                                       C.c_void_p]),
    ("mn_graph_last_ms", C.c_double, [C.c_void_p]),
    ("mn_graph_out_edge_count", C.c_longlong, [C.c_void_p]),
    ("mn_graph_out_lists", C.c_int, [C.c_void_p, _i32p, _i32p]),
    ("mn_graph_pagerank", C.c_int, [C.c_int, C.c_int64, _i32p, _i32p, C.c_double, C.c_int, C.c_int,
                                    np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.POINTER(AlgoStats)]),
    ("mn_graph_components", C.c_int, [C.c_int, C.c_int64, _i32p, _i32p, C.c_int, C.c_int, _i32p, _i32p, C.POINTER(AlgoStats)]),
    ("mn_graph_algo_last_error", C.c_char_p, []),
    ("mn_csr_apply_delta", C.c_int, [C.c_int, _i32p, _i32p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, _i32p,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    ("mn_host_free", None, [C.c_void_p]),
]


def _glib():
    L = lib()
    if not getattr(L, "_graph_bound", False):
        for name, res, args in GRAPH_SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        L._graph_bound = True
    return L


def _gerr():
    m = _glib().mn_graph_last_error()
    return m.decode() if m else ""


class CsrBlock(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("offsets_bytes", C.c_int), ("targets", C.c_void_p), ("targets_bytes", C.c_int),
                ("weights", C.c_void_p), ("weights_bytes", C.c_int)]


class Graph:
    """Device copy of GraphData.out / GraphData.in as CSR.  off/tgt int32, weights float64 or None."""

    def __init__(self, n, off_out, tgt_out, w_out, off_in, tgt_in, w_in, device=0):
        self.L = _glib()
        self.n = int(n)
        keep = [np.ascontiguousarray(off_out, np.int32), np.ascontiguousarray(tgt_out, np.int32),
                None if w_out is None else np.ascontiguousarray(w_out, np.float64),
                np.ascontiguousarray(off_in, np.int32), np.ascontiguousarray(tgt_in, np.int32),
                None if w_in is None else np.ascontiguousarray(w_in, np.float64)]
        ptr = lambda a: None if a is None else a.ctypes.data
        self.h = self.L.mn_graph_create(self.n, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]), ptr(keep[5]),
                                        device)
        if not self.h:
            raise MuninnHipError("mn_graph_create failed: " + _gerr())

    @classmethod
    def from_blocks(cls, n, fwd, rev, device=0):
        """fwd / rev: lists of (offsets bytes, targets bytes, weights bytes | None) — the rows of the reference's
        "{t}_csr_fwd" / "{t}_csr_rev" shadow tables in block_id order (mn_graph_create_blocked)."""
        self = cls.__new__(cls)
        self.L = _glib()
        self.n = int(n)

        def pack(rows):
            arr = (CsrBlock * max(1, len(rows)))()
            keep = []
            for i, (o, t, w) in enumerate(rows):
                bufs = [C.create_string_buffer(bytes(x), len(x)) if x else None for x in (o, t, w)]
                keep.append(bufs)
                arr[i] = CsrBlock(C.cast(bufs[0], C.c_void_p) if bufs[0] else None, len(o) if o else 0,
                                  C.cast(bufs[1], C.c_void_p) if bufs[1] else None, len(t) if t else 0,
                                  C.cast(bufs[2], C.c_void_p) if bufs[2] else None, len(w) if w else 0)
            return arr, keep

        fa, k1 = pack(fwd)
        ra, k2 = pack(rev)
        self.h = self.L.mn_graph_create_blocked(self.n, C.byref(fa), len(fwd), C.byref(ra), len(rev), device)
        if not self.h:
            raise MuninnHipError("mn_graph_create_blocked failed: " + _gerr())
        return self

    def close(self):
        if getattr(self, "h", None):
            self.L.mn_graph_destroy(self.h)
            self.h = None

    __del__ = close

    def betweenness(self, direction="forward", auto_approx=0, normalized=0, edges=False):
        """brandes_compute (src/graph_centrality.c:393-505) → (cb[n], eb[n][n] or None, device ms)"""
        cb = np.zeros(max(self.n, 1), np.float64)
        eb = np.zeros((self.n, self.n), np.float64) if edges else None
        d = {"both": 0, "forward": 1, "reverse": 2}[direction]
        if self.L.mn_graph_betweenness(self.h, d, int(auto_approx), int(normalized), cb, eb.ctypes.data if edges else None) != 0:
            raise MuninnHipError(_gerr())
        return cb[:self.n], eb, self.L.mn_graph_last_ms(self.h)

    def leiden(self, resolution=1.0, direction="both", mode=LEIDEN_SEQUENTIAL, batch=0):
        """run_leiden → (community[n] int32, Q, stats dict)"""
        comm = np.empty(max(self.n, 1), np.int32)
        q = C.c_double(0.0)
        rc = self.L.mn_graph_leiden(self.h, float(resolution), 1 if direction == "both" else 0, mode, batch, comm, C.byref(q))
        if rc != 0:
            raise MuninnHipError(_gerr())
        st = LeidenStats()
        self.L.mn_graph_leiden_stats(self.h, C.byref(st))
        return comm[:self.n], q.value, {n: getattr(st, n) for n, _ in LeidenStats._fields_}


def node2vec_train(off, adj, dim, p=1.0, q=1.0, num_walks=10, walk_length=80, window=5, neg_samples=5, learning_rate=0.025,
                   epochs=1, mode=N2V_SEQUENTIAL, device=0, batch_walks=0):
    """node2vec_train's compute (src/node2vec.c:486-551) on the device → (embeddings [n][dim] f32, stats)."""
    L = _glib()
    off = np.ascontiguousarray(off, np.int32)
    adj = np.ascontiguousarray(adj if len(adj) else np.zeros(1, np.int32), np.int32)
    n = len(off) - 1
    out = np.zeros((max(n, 1), dim), np.float32)
    prm = N2vParams(dim, p, q, num_walks, walk_length, window, neg_samples, learning_rate, epochs, batch_walks)
    st = N2vStats()
    rc = L.mn_node2vec_train(n, off, adj, C.byref(prm), mode, device, out, C.byref(st))
    if rc < 0:
        m = L.mn_node2vec_last_error()
        raise MuninnHipError(m.decode() if m else "mn_node2vec_train failed")
    return out[:n], {"pairs": st.pairs, "device_ms": st.device_ms}


# ───────────── host-side CSR builders (the device input formats; node ids are already indices) ─────────────

def csr_pair_from_edges(n, src, dst, weights=None):
    """GraphData.out / GraphData.in as CSR (src/graph_csr.c:20-70 over src/graph_load.c:218-246): out[src] lists dst
    and in[dst] lists src, each in edge order.  → (off_out, tgt_out, w_out, off_in, tgt_in, w_in); weights None ⇒ None."""
    src = np.asarray(src, np.int64)
    dst = np.asarray(dst, np.int64)
    w = None if weights is None else np.asarray(weights, np.float64)

    def one(keys, vals):
        o = np.argsort(keys, kind="stable")
        off = np.zeros(n + 1, np.int64)
        np.add.at(off, keys + 1, 1)
        return np.cumsum(off).astype(np.int32), vals[o].astype(np.int32), None if w is None else w[o].copy()

    return one(src, dst) + one(dst, src)


def graph_from_edges(n, src, dst, weights=None, device=0):
    """Graph (Leiden input) from an edge list with integer node indices."""
    return Graph(n, *csr_pair_from_edges(n, src, dst, weights), device=device)


def node2vec_train_into(off, adj, dim, index, first_rowid=1, want_embeddings=True, p=1.0, q=1.0, num_walks=10, walk_length=80,
                        window=5, neg_samples=5, learning_rate=0.025, epochs=1, batch_walks=0):
    """mn_node2vec_train_into: train (MN_N2V_BATCHED) on the index's device and build `index` (an hnsw.HnswIndex) from the
    embeddings without a host round trip → (embeddings [n][dim] f32 or None, stats with 'build_seconds')."""
    L = _glib()
    off = np.ascontiguousarray(off, np.int32)
    adj = np.ascontiguousarray(adj if len(adj) else np.zeros(1, np.int32), np.int32)
    n = len(off) - 1
    out = np.zeros((max(n, 1), dim), np.float32) if want_embeddings else None
    prm = N2vParams(dim, p, q, num_walks, walk_length, window, neg_samples, learning_rate, epochs, batch_walks)
    st = N2vStats()
    bs = C.c_double(0.0)
    rc = L.mn_node2vec_train_into(n, off, adj, C.byref(prm), N2V_BATCHED, index.h, int(first_rowid),
                                  out.ctypes.data if out is not None else None, C.byref(st), C.byref(bs))
    if rc < 0:
        raise MuninnHipError((L.mn_node2vec_last_error() or b"").decode())
    return (out[:n] if out is not None else None), {"pairs": st.pairs, "device_ms": st.device_ms, "build_seconds": bs.value}


def n2v_csr_from_edges(n, src, dst):
    """node2vec.c's own adjacency (src/node2vec.c:96-134): undirected, duplicate edges dropped, every list in
    insertion order (edge i adds src→dst, then dst→src).  → (off int32[n+1], adj int32[E])."""
    src = np.asarray(src, np.int64)
    dst = np.asarray(dst, np.int64)
    a = np.empty(2 * len(src), np.int64)
    b = np.empty(2 * len(src), np.int64)
    a[0::2], a[1::2] = src, dst
    b[0::2], b[1::2] = dst, src
    _, first = np.unique(a * n + b, return_index=True)
    first.sort()
    a, b = a[first], b[first]
    o = np.argsort(a, kind="stable")
    off = np.zeros(n + 1, np.int64)
    np.add.at(off, a + 1, 1)
    return np.cumsum(off).astype(np.int32), b[o].astype(np.int32)


# ───────────── f-4: graph_tvf.c's edge-list algorithms (nodes = first-seen indices, edges in row order) ─────────────

def _edges(src, dst):
    src, dst = np.ascontiguousarray(src, np.int32), np.ascontiguousarray(dst, np.int32)
    pad = np.zeros(1, np.int32)
    return (src if len(src) else pad), (dst if len(dst) else pad), len(src)


def pagerank(n, src, dst, damping=0.85, iterations=20, device=0):
    """run_pagerank (src/graph_tvf.c:1631-1797) → (rank[n] float64, stats)"""
    L = _glib()
    s, d, ne = _edges(src, dst)
    out = np.zeros(max(n, 1), np.float64)
    st = AlgoStats()
    if L.mn_graph_pagerank(n, ne, s, d, float(damping), int(iterations), device, out, C.byref(st)) != 0:
        raise MuninnHipError((L.mn_graph_algo_last_error() or b"").decode())
    return out[:n], {"device_ms": st.device_ms, "iterations": st.iterations, "dangling": st.aux}


def components(n, src, dst, mode=COMPONENTS_EXACT, device=0):
    """run_components (src/graph_tvf.c:1314-1366) → (component_id[n], component_size[n], stats)"""
    L = _glib()
    s, d, ne = _edges(src, dst)
    cid, csz = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    st = AlgoStats()
    if L.mn_graph_components(n, ne, s, d, mode, device, cid, csz, C.byref(st)) != 0:
        raise MuninnHipError((L.mn_graph_algo_last_error() or b"").decode())
    return cid[:n], csz[:n], {"device_ms": st.device_ms, "rounds": st.iterations}


class CsrDelta(C.Structure):  # CsrDelta, src/graph_csr.h:37-42
    _fields_ = [("src_idx", C.c_int32), ("dst_idx", C.c_int32), ("weight", C.c_double), ("op", C.c_int)]


def csr_apply_delta(off, tgt, w, dsrc, ddst, dw, dop, new_n, device=0):
    """csr_apply_delta (src/graph_csr.c:175-325) on the device → (new_off, new_tgt, new_w or None)"""
    L = _glib()
    off = np.ascontiguousarray(off, np.int32)
    old_n = len(off) - 1
    tgt = np.ascontiguousarray(tgt if len(tgt) else np.zeros(1, np.int32), np.int32)
    wv = None if w is None else np.ascontiguousarray(w if len(w) else np.zeros(1), np.float64)
    nd = len(dsrc)
    dl = (CsrDelta * max(nd, 1))()
    for i in range(nd):
        dl[i] = CsrDelta(int(dsrc[i]), int(ddst[i]), float(dw[i]), int(dop[i]))
    new_off = np.zeros(max(new_n, old_n) + 1, np.int32)
    pt, pw, ne = C.c_void_p(), C.c_void_p(), C.c_int(0)
    rc = L.mn_csr_apply_delta(old_n, off, tgt, None if wv is None else wv.ctypes.data, 0 if wv is None else 1, dl, nd, new_n, device,
                              new_off, C.byref(pt), C.byref(pw), C.byref(ne))
    if rc != 0:
        raise MuninnHipError((L.mn_graph_algo_last_error() or b"").decode())
    e = ne.value
    new_tgt = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_int32)), (e,)).copy() if e else np.zeros(0, np.int32)
    new_w = None
    if wv is not None:
        new_w = np.ctypeslib.as_array(C.cast(pw, C.POINTER(C.c_double)), (e,)).copy() if e else np.zeros(0, np.float64)
    L.mn_host_free(pt)
    L.mn_host_free(pw)
    return new_off, new_tgt, new_w
