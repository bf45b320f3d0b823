"""Exact (reference-semantics) inserts into an existing index: vec/s and, with MN_SPEC_TRACE=1, the speculative rounds.
usage: probe_spec.py N DIM [inserts] [metric]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
N, D = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
metric = sys.argv[4] if len(sys.argv) > 4 else "cosine"
rng = np.random.default_rng(42)
X = rng.standard_normal((N + K, D), dtype=np.float32)
g = pkg.HnswIndex(D, metric, 16, 200)
if N:
    assert g.build(np.arange(1, N + 1, dtype=np.int64), X[:N], 16, 8192) == 0
    g.sync()
for rep in range(2):
    a = N + rep * (K // 2)
    ids = np.arange(a + 1, a + 1 + K // 2, dtype=np.int64)
    t0 = time.perf_counter()
    assert g.insert_batch(ids, X[a:a + K // 2], pkg.BUILD_SEQUENTIAL) == 0
    g.sync()
    dt = time.perf_counter() - t0
    print(f"N={a} D={D}: {K // 2} exact inserts in {dt:.2f}s = {K // 2 / dt:.0f} vec/s", flush=True)
g.close()
