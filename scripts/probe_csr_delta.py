#!/usr/bin/env python3
"""f-2 timing: mn_csr_apply_delta (host arrays in, host arrays out — the boundary src/graph_adjacency.c:864,910 calls through)
beside the reference's own csr_apply_delta (src/graph_csr.c:175, compiled into oracle/_ref) on the same CSR and delta log.
Two shapes: one 4 096-node block as graph_adjacency stores them, and a whole 1M-node / 20M-edge CSR with a 1M-entry log.
Prints one JSON line; results are compared (offsets, targets) before any time is reported."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import muninn_amd  # noqa: E402,F401
import sqlite_muninn_amd.graph as mng  # noqa: E402
from oracle import orc_graph as og  # noqa: E402

DELTA = np.dtype({"names": ["src", "dst", "w", "op"], "formats": [np.int32, np.int32, np.float64, np.int32],
                  "offsets": [0, 4, 8, 16], "itemsize": 24})


def case(seed, n, e, nd):
    r = np.random.default_rng(seed)
    src = np.sort(r.integers(0, n, e)).astype(np.int32)
    tgt = r.integers(0, n, e).astype(np.int32)
    off = np.zeros(n + 1, np.int64)
    np.add.at(off, src.astype(np.int64) + 1, 1)
    off = np.cumsum(off).astype(np.int32)
    dsrc = r.integers(0, n, nd).astype(np.int32)
    ddst = r.integers(0, n, nd).astype(np.int32)
    dop = r.choice([1, 2], nd).astype(np.int32)
    pick = r.integers(0, e, nd)
    hit = dop == 2
    dsrc[hit], ddst[hit] = src[pick][hit], tgt[pick][hit]
    return off, tgt, dsrc, ddst, r.random(nd), dop


def device(L, off, tgt, dsrc, ddst, dw, dop, reps):
    n, nd = len(off) - 1, len(dsrc)
    dl = np.zeros(nd, DELTA)
    dl["src"], dl["dst"], dl["w"], dl["op"] = dsrc, ddst, dw, dop
    new_off = np.zeros(n + 1, np.int32)
    best, out = 1e9, None
    for _ in range(reps):
        pt, pw, ne = C.c_void_p(), C.c_void_p(), C.c_int(0)
        t0 = time.perf_counter()
        rc = L.mn_csr_apply_delta(n, off, tgt, None, 0, C.cast(dl.ctypes.data, C.c_void_p), nd, n, 0, new_off, C.byref(pt), C.byref(pw),
                                  C.byref(ne))
        dt = time.perf_counter() - t0
        assert rc == 0
        out = (new_off.copy(), np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_int32)), (ne.value,)).copy())
        L.mn_host_free(pt)
        L.mn_host_free(pw)
        best = min(best, dt)
    return best, out


def main():
    L = mng._glib()
    res = {}
    for name, (n, e, nd, reps) in {"block_4096_nodes_80k_edges_2k_deltas": (4096, 80_000, 2000, 20),
                                    "whole_1M_nodes_20M_edges_1M_deltas": (1_000_000, 20_000_000, 1_000_000, 3)}.items():
        off, tgt, dsrc, ddst, dw, dop = case(7, n, e, nd)
        device(L, off, tgt, dsrc, ddst, dw, dop, 1)  # code objects, first allocations
        tg, got = device(L, off, tgt, dsrc, ddst, dw, dop, reps)
        fn = C.CDLL(og.REF_EXT_SO).ref_csr_apply_delta
        i32p, f64p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"), np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        fn.argtypes = [C.c_int, i32p, i32p, f64p, C.c_int, C.c_int, i32p, i32p, f64p, i32p, C.c_int, i32p, i32p, f64p]
        ro, rt, rw, tr = np.zeros(n + 1, np.int32), np.zeros(e + nd + 1, np.int32), np.zeros(e + nd + 1), 1e9
        for _ in range(max(1, reps // 2)):
            t0 = time.perf_counter()
            ne = fn(n, off, tgt, np.zeros(1), 0, nd, dsrc, ddst, dw, dop, n, ro, rt, rw)
            tr = min(tr, time.perf_counter() - t0)
        rt = rt[:ne]
        same = bool(np.array_equal(got[0], ro) and np.array_equal(got[1], rt))
        res[name] = {"device_ms_host_to_host": round(tg * 1e3, 3), "reference_cpu_ms": round(tr * 1e3, 3), "same_csr": same}
    print(json.dumps({"probe": "csr_apply_delta", **res}))


if __name__ == "__main__":
    main()
