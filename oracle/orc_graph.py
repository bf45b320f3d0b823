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
