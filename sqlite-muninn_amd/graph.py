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


GRAPH_SYMBOLS = [
    ("mn_graph_create", C.c_void_p, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("mn_graph_destroy", None, [C.c_void_p]),
    ("mn_graph_last_error", C.c_char_p, []),
    ("mn_graph_leiden", C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, _i32p, C.POINTER(C.c_double)]),
    ("mn_graph_leiden_stats", C.c_int, [C.c_void_p, C.POINTER(LeidenStats)]),
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

    def close(self):
        if getattr(self, "h", None):
            self.L.mn_graph_destroy(self.h)
            self.h = None

    __del__ = close

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
