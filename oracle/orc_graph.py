"""ctypes bindings for the graph half of the CPU oracle (Leiden; Node2Vec) and for the compiled
reference's run_leiden (oracle/_ref/muninn.so + ref_graph_access.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import orc

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
REF_EXT_SO = orc.REF_EXT + ".so"


class _Graph(C.Structure):
    _fields_ = [("n", C.c_int), ("off_out", C.c_void_p), ("tgt_out", C.c_void_p), ("w_out", C.c_void_p),
                ("off_in", C.c_void_p), ("tgt_in", C.c_void_p), ("w_in", C.c_void_p)]


class _LeidenStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("iterations", "moves", "move_sweeps", "refine_sweeps")]


class Csr:
    """GraphData (src/graph_load.h:27-37) as two CSR arrays, built exactly as graph_data_load does
    (src/graph_load.c:218-246): node index = first appearance scanning rows (src then dst);
    out[src] gets (dst, w) when forward is loaded, in[dst] gets (src, w) when reverse is loaded."""

    def __init__(self, src, dst, w=None, direction="both", n_nodes=None, first_seen=True):
        src = np.asarray(src, np.int64)
        dst = np.asarray(dst, np.int64)
        if first_seen:
            inter = np.empty(2 * len(src), np.int64)
            inter[0::2], inter[1::2] = src, dst
            uniq, first = np.unique(inter, return_index=True)
            order = np.argsort(first, kind="stable")
            self.node_ids = uniq[order]  # node index -> original id
            remap = np.empty(len(uniq), np.int64)
            remap[order] = np.arange(len(uniq))
            s = remap[np.searchsorted(uniq, src)]
            d = remap[np.searchsorted(uniq, dst)]
            n = len(uniq)
        else:
            s, d = src, dst
            n = int(n_nodes)
            self.node_ids = np.arange(n)
        self.n = n
        self.direction = direction
        self.weighted = w is not None
        ww = np.ones(len(s), np.float64) if w is None else np.asarray(w, np.float64)
        fwd, rev = direction != "reverse", direction != "forward"

        def build(keys, vals, use):
            if not use:
                return np.zeros(n + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float64)
            o = np.argsort(keys, kind="stable")
            off = np.zeros(n + 1, np.int64)
            np.add.at(off, keys + 1, 1)
            off = np.cumsum(off).astype(np.int32)
            return off, vals[o].astype(np.int32), ww[o].copy()

        self.off_out, self.tgt_out, self.w_out = build(s, d, fwd)
        self.off_in, self.tgt_in, self.w_in = build(d, s, rev)
        self.src_idx, self.dst_idx, self.w = s.astype(np.int32), d.astype(np.int32), ww

    def c_struct(self):
        g = _Graph()
        g.n = self.n
        g.off_out, g.tgt_out = self.off_out.ctypes.data, self.tgt_out.ctypes.data
        g.w_out = self.w_out.ctypes.data if self.weighted else None
        g.off_in, g.tgt_in = self.off_in.ctypes.data, self.tgt_in.ctypes.data
        g.w_in = self.w_in.ctypes.data if self.weighted else None
        return g


def _lib():
    L = orc.lib()
    if not getattr(L, "_graph_bound", False):
        L.orc_leiden.restype = C.c_double
        L.orc_leiden.argtypes = [C.POINTER(_Graph), _i32p, C.c_double, C.c_int, C.c_int, C.POINTER(_LeidenStats)]
        L.orc_modularity.restype = C.c_double
        L.orc_modularity.argtypes = [C.POINTER(_Graph), _i32p, C.c_double, C.c_double, C.c_int]
        L._graph_bound = True
    return L


def leiden(csr: Csr, resolution=1.0, batch=1):
    """orc_leiden → (community[n], Q, stats)"""
    g = csr.c_struct()
    comm = np.empty(csr.n, np.int32)
    st = _LeidenStats()
    q = _lib().orc_leiden(C.byref(g), comm, float(resolution), 1 if csr.direction == "both" else 0, int(batch), C.byref(st))
    return comm, q, {n: getattr(st, n) for n, _ in _LeidenStats._fields_}


def have_ref_graph() -> bool:
    return os.path.exists(REF_EXT_SO)


_ref = None


def ref_leiden(src, dst, w=None, direction="both", resolution=1.0):
    """The reference's own run_leiden on the reference's own GraphData (build container only).
    Node ids must be 0..n_ids-1; returns (community per reference node index, Q, index_of_id)."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_EXT_SO)
        _ref.ref_leiden_edges.restype = C.c_double
        _ref.ref_leiden_edges.argtypes = [C.c_int, C.c_int, _i32p, _i32p, C.c_void_p, C.c_int, C.c_double, _i32p, _i32p,
                                          C.POINTER(C.c_int)]
    src = np.ascontiguousarray(src, np.int32)
    dst = np.ascontiguousarray(dst, np.int32)
    n_ids = int(max(src.max(), dst.max())) + 1 if len(src) else 0
    wv = None if w is None else np.ascontiguousarray(w, np.float64)
    idx = np.empty(max(n_ids, 1), np.int32)
    comm = np.empty(max(n_ids, 1), np.int32)
    nn = C.c_int(0)
    d = {"both": 0, "forward": 1, "reverse": 2}[direction]
    q = _ref.ref_leiden_edges(n_ids, len(src), src, dst, wv.ctypes.data if wv is not None else None, d, float(resolution), idx,
                              comm, C.byref(nn))
    return comm[:nn.value].copy(), q, idx[:n_ids].copy()


# ───────────────────────── Node2Vec ─────────────────────────

class _N2vGraph(C.Structure):
    _fields_ = [("n", C.c_int), ("off", C.c_void_p), ("adj", C.c_void_p)]


class _N2vParams(C.Structure):
    _fields_ = [("dim", C.c_int), ("p", C.c_double), ("q", C.c_double), ("num_walks", C.c_int), ("walk_length", C.c_int),
                ("window", C.c_int), ("neg_samples", C.c_int), ("lr", C.c_double), ("epochs", C.c_int)]


def _n2v_lib():
    L = _lib()
    if not getattr(L, "_n2v_bound", False):
        L.orc_n2v_build_graph.argtypes = [C.c_int, _i32p, _i32p, C.c_int, _i32p, _i32p, _i32p]
        L.orc_node2vec_train.argtypes = [C.POINTER(_N2vGraph), C.POINTER(_N2vParams), np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"),
                                         C.POINTER(C.c_int64)]
        L.orc_node2vec_train_batched.argtypes = [C.POINTER(_N2vGraph), C.POINTER(_N2vParams), C.c_int,
                                                 np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.POINTER(C.c_int64)]
        L.orc_biased_walk.argtypes = [C.POINTER(_N2vGraph), C.c_int, C.c_double, C.c_double, C.c_int, _i32p, C.POINTER(C.c_uint)]
        L.orc_n2v_sigmoid_table.restype = C.POINTER(C.c_float)
        L.orc_n2v_neg_table.argtypes = [C.POINTER(_N2vGraph), _i32p]
        L._n2v_bound = True
    return L


class N2vGraph:
    """node2vec.c's Graph (first-seen node order, undirected, de-duplicated) as CSR."""

    def __init__(self, src, dst):
        src = np.ascontiguousarray(src, np.int32)
        dst = np.ascontiguousarray(dst, np.int32)
        n_ids = int(max(src.max(), dst.max())) + 1 if len(src) else 0
        off = np.zeros(n_ids + 2, np.int32)
        adj = np.zeros(max(1, 2 * len(src)), np.int32)
        idx = np.full(max(1, n_ids), -1, np.int32)
        self.n = _n2v_lib().orc_n2v_build_graph(len(src), src, dst, n_ids, off, adj, idx)
        self.off = off[:self.n + 1].copy()
        self.adj = adj[:self.off[-1]].copy() if self.n else np.zeros(0, np.int32)
        self.index_of_id = idx[:n_ids]

    def c_struct(self):
        g = _N2vGraph()
        g.n = self.n
        g.off = self.off.ctypes.data
        g.adj = self.adj.ctypes.data
        return g


def node2vec_train(g: N2vGraph, dim, p, q, num_walks, walk_length, window, neg, lr, epochs):
    """→ (embeddings [n][dim] f32 L2-normalised, number of (center, context) pairs)"""
    out = np.zeros((max(g.n, 1), dim), np.float32)
    prm = _N2vParams(dim, p, q, num_walks, walk_length, window, neg, lr, epochs)
    cg = g.c_struct()
    npairs = C.c_int64(0)
    _n2v_lib().orc_node2vec_train(C.byref(cg), C.byref(prm), out, C.byref(npairs))
    return out[:g.n], npairs.value


def node2vec_train_batched(g: N2vGraph, dim, p, q, num_walks, walk_length, window, neg, lr, epochs, batch):
    out = np.zeros((max(g.n, 1), dim), np.float32)
    prm = _N2vParams(dim, p, q, num_walks, walk_length, window, neg, lr, epochs)
    cg = g.c_struct()
    npairs = C.c_int64(0)
    _n2v_lib().orc_node2vec_train_batched(C.byref(cg), C.byref(prm), int(batch), out, C.byref(npairs))
    return out[:g.n], npairs.value


def biased_walk(g: N2vGraph, start, p, q, walk_length, rng_state):
    walk = np.zeros(walk_length, np.int32)
    st = C.c_uint(rng_state)
    cg = g.c_struct()
    n = _n2v_lib().orc_biased_walk(C.byref(cg), start, p, q, walk_length, walk, C.byref(st))
    return walk[:n].copy(), st.value


def ref_node2vec_sql(edges, dim, p, q, num_walks, walk_length, window, neg, lr, epochs):
    """The reference's node2vec_train through its own SQL surface (oracle/_ref/muninn.so): returns the
    embedding bytes it INSERTs (read back from the output table's _nodes shadow table), in rowid order."""
    import sqlite3

    c = sqlite3.connect(":memory:")
    c.enable_load_extension(True)
    c.load_extension(orc.REF_EXT)
    c.execute("CREATE TABLE e (src TEXT, dst TEXT)")
    c.executemany("INSERT INTO e VALUES (?, ?)", [(str(a), str(b)) for a, b in edges])
    c.execute(f"CREATE VIRTUAL TABLE emb USING hnsw_index(dimensions={dim}, metric='cosine', m=8, ef_construction=50)")
    n = c.execute("SELECT node2vec_train('e', 'src', 'dst', 'emb', ?, ?, ?, ?, ?, ?, ?, ?, ?)",
                  (dim, p, q, num_walks, walk_length, window, neg, lr, epochs)).fetchone()[0]
    rows = c.execute("SELECT id, vector FROM emb_nodes ORDER BY id").fetchall()
    c.close()
    assert n == len(rows)
    return np.array([np.frombuffer(r[1], np.float32) for r in rows], np.float32).reshape(n, dim)


# ───────────────────────── f-4: PageRank, components ─────────────────────────

def first_seen_edges(rows):
    """rows: iterable of (src, dst) node ids as the edge table holds them (None = SQL NULL, skipped as the reference
    skips it).  Returns (ids in first-seen order — src of a row before its dst — , src idx, dst idx)."""
    idx, ids, s, d = {}, [], [], []
    for a, b in rows:
        if a is None or b is None:
            continue
        for x in (a, b):
            if x not in idx:
                idx[x] = len(ids)
                ids.append(x)
        s.append(idx[a])
        d.append(idx[b])
    return ids, np.asarray(s, np.int32), np.asarray(d, np.int32)


def pagerank(n, src, dst, damping=0.85, iterations=20):
    L = _lib()
    L.orc_pagerank.argtypes = [C.c_int, C.c_int, _i32p, _i32p, C.c_double, C.c_int, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
    src, dst = np.ascontiguousarray(src, np.int32), np.ascontiguousarray(dst, np.int32)
    out = np.zeros(max(n, 1), np.float64)
    assert L.orc_pagerank(n, len(src), src if len(src) else np.zeros(1, np.int32), dst if len(dst) else np.zeros(1, np.int32),
                          float(damping), int(iterations), out) == 0
    return out[:n]


def components(n, src, dst):
    L = _lib()
    L.orc_components.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p]
    src, dst = np.ascontiguousarray(src, np.int32), np.ascontiguousarray(dst, np.int32)
    cid, csz = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    assert L.orc_components(n, len(src), src if len(src) else np.zeros(1, np.int32), dst if len(dst) else np.zeros(1, np.int32),
                            cid, csz) == 0
    return cid[:n], csz[:n]


def ref_graph_tvf(rows, damping=None, iterations=None):
    """The reference's own graph_pagerank / graph_components through its SQL surface (oracle/_ref/muninn.so, build
    container only).  rows: (src, dst) text ids or None.  Returns {"pagerank": [(node, rank)], "components": [(node, id, size)]}
    in the order the TVFs emit them."""
    import sqlite3

    c = sqlite3.connect(":memory:")
    c.enable_load_extension(True)
    c.load_extension(REF_EXT_SO[:-3])
    c.execute("CREATE TABLE e(s TEXT, d TEXT)")
    c.executemany("INSERT INTO e VALUES (?, ?)", list(rows))
    extra, args = "", []
    if damping is not None:
        extra += " AND damping = ?"
        args.append(damping)
    if iterations is not None:
        extra += " AND iterations = ?"
        args.append(iterations)
    pr = c.execute("SELECT node, rank FROM graph_pagerank WHERE edge_table='e' AND src_col='s' AND dst_col='d'" + extra, args).fetchall()
    cc = c.execute("SELECT node, component_id, component_size FROM graph_components WHERE edge_table='e' AND src_col='s' AND dst_col='d'").fetchall()
    c.close()
    return {"pagerank": pr, "components": cc}


# ───────────────────────── f-2: csr_apply_delta ─────────────────────────

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def _delta_call(fn, off, tgt, w, dsrc, ddst, dw, dop, new_n):
    off = np.ascontiguousarray(off, np.int32)
    old_n = len(off) - 1
    tgt = np.ascontiguousarray(tgt if len(tgt) else np.zeros(1, np.int32), np.int32)
    has_w = w is not None
    wv = np.ascontiguousarray(w if has_w and len(w) else np.zeros(1), np.float64)
    nd = len(dsrc)
    pad = lambda a, t: np.ascontiguousarray(a if len(a) else np.zeros(1, t), t)  # noqa: E731
    n_out = max(new_n, old_n)
    new_off = np.zeros(n_out + 1, np.int32)
    cap = int(off[-1]) + nd + 1
    new_tgt, new_w = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
    fn.argtypes = [C.c_int, _i32p, _i32p, _f64p, C.c_int, C.c_int, _i32p, _i32p, _f64p, _i32p, C.c_int, _i32p, _i32p, _f64p]
    e = fn(old_n, off, tgt, wv, 1 if has_w else 0, nd, pad(dsrc, np.int32), pad(ddst, np.int32), pad(dw, np.float64),
           pad(dop, np.int32), new_n, new_off, new_tgt, new_w)
    assert e >= 0
    return new_off, new_tgt[:e].copy(), (new_w[:e].copy() if has_w else None)


def csr_apply_delta(off, tgt, w, dsrc, ddst, dw, dop, new_n):
    return _delta_call(_lib().orc_csr_apply_delta, off, tgt, w, dsrc, ddst, dw, dop, new_n)


def ref_csr_apply_delta(off, tgt, w, dsrc, ddst, dw, dop, new_n):
    """the reference's own csr_apply_delta (oracle/_ref/muninn.so, build container only)"""
    return _delta_call(C.CDLL(REF_EXT_SO).ref_csr_apply_delta, off, tgt, w, dsrc, ddst, dw, dop, new_n)


def delta_case(seed, n=300, e=2000, nd=900, weighted=False, new_nodes=40):
    """A random CSR (multi-edges included) and a delta log: inserts, deletes of present / absent / repeated edges,
    edges to and from nodes beyond the old node count, out-of-range rows."""
    r = np.random.default_rng(seed)
    src = np.sort(r.integers(0, n, e)).astype(np.int32)
    tgt = r.integers(0, n, e).astype(np.int32)
    off = np.zeros(n + 1, np.int32)
    np.add.at(off, src + 1, 1)
    off = np.cumsum(off).astype(np.int32)
    w = r.random(e) if weighted else None
    new_n = n + new_nodes
    dsrc = r.integers(-2, new_n + 2, nd).astype(np.int32)
    ddst = r.integers(-2, new_n + 2, nd).astype(np.int32)
    dop = r.choice([1, 2, 2, 3], nd).astype(np.int32)
    pick = r.integers(0, e, nd // 2)  # make half of the deletes hit existing edges
    dsrc[: nd // 2][dop[: nd // 2] == 2] = src[pick][dop[: nd // 2] == 2]
    ddst[: nd // 2][dop[: nd // 2] == 2] = tgt[pick][dop[: nd // 2] == 2]
    dw = r.random(nd)
    return off, tgt, w, dsrc, ddst, dw, dop, new_n


# ───────────────────────── f-4: Brandes betweenness ─────────────────────────

_DIR = {"both": 0, "forward": 1, "reverse": 2}


def betweenness(csr: Csr, auto_approx=0, normalized=0, edges=False):
    """orc_betweenness on a Csr built for the traversal direction → (cb[n], eb[n][n] or None)"""
    L = _lib()
    L.orc_betweenness.argtypes = [C.POINTER(_Graph), C.c_int, C.c_int, C.c_int, _f64p, C.c_void_p]
    g = csr.c_struct()
    cb = np.zeros(max(csr.n, 1), np.float64)
    eb = np.zeros((csr.n, csr.n), np.float64) if edges else None
    assert L.orc_betweenness(C.byref(g), _DIR[csr.direction], int(auto_approx), int(normalized), cb,
                             eb.ctypes.data if edges else None) == 0
    return cb[:csr.n], eb


def ref_betweenness_sql(rows, weighted=False, direction=None, normalized=None, auto_approx=None):
    """The reference's graph_node_betweenness / graph_edge_betweenness through its SQL surface (build container only).
    rows: (src, dst[, weight]).  Returns ([(node, centrality)], [(src, dst, centrality)])."""
    import sqlite3

    c = sqlite3.connect(":memory:")
    c.enable_load_extension(True)
    c.load_extension(REF_EXT_SO[:-3])
    c.execute("CREATE TABLE e(s TEXT, d TEXT, w REAL)")
    c.executemany("INSERT INTO e VALUES (?, ?, ?)", [(r[0], r[1], r[2] if len(r) > 2 else None) for r in rows])
    extra, args = "", []
    for col, val in (("weight_col", "w" if weighted else None), ("direction", direction), ("normalized", normalized),
                     ("auto_approx_threshold", auto_approx)):
        if val is not None:
            extra += f" AND {col} = ?"
            args.append(val)
    where = "edge_table='e' AND src_col='s' AND dst_col='d'" + extra
    nodes = c.execute("SELECT node, centrality FROM graph_node_betweenness WHERE " + where, args).fetchall()
    edges = c.execute("SELECT src, dst, centrality FROM graph_edge_betweenness WHERE " + where, args).fetchall()
    c.close()
    return nodes, edges
