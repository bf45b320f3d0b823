"""single-query latency A/B helper: 1M x 768 index, one mn_hnsw_search at a time (the SQL surface's xFilter shape)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
N, D = 1000000, 768
X = np.random.default_rng(42).standard_normal((N, D), dtype=np.float32)
Q = np.random.default_rng(43).standard_normal((300, D), dtype=np.float32)
g = pkg.HnswIndex(D, "cosine", 16, 200)
g.build(np.arange(1, N + 1, dtype=np.int64), X)
for ef in (128, 64):
    ts = []
    for i in range(300):
        t = time.perf_counter(); g.search(Q[i], 10, ef); ts.append((time.perf_counter() - t) * 1e3)
    print(f"single query ef={ef}: median {np.median(ts[50:]):.3f} ms  p10 {np.percentile(ts[50:], 10):.3f}  p90 {np.percentile(ts[50:], 90):.3f}", flush=True)
