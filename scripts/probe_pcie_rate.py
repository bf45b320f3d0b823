"""kNN rate when queries come from and results go to HOST buffers (mn_hnsw_search_batch) vs HBM-resident (bench value)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
from bench import gen_vectors
pkg = muninn_amd.pkg
N, D, NQ = 1_000_000, 768, 10000
X = gen_vectors(N, D, 42, "gaussian"); Q = gen_vectors(NQ, D, 43, "gaussian")
g = pkg.HnswIndex(D, "cosine", 16, 200)
g.build(np.arange(1, N + 1, dtype=np.int64), X, 16, 8192); g.sync()
g.search_batch(Q, 10, 128)
t = time.perf_counter()
for _ in range(10): g.search_batch(Q, 10, 128)
host = (time.perf_counter() - t) / 10
dq = g.dev_malloc(Q.nbytes); g.dev_upload(dq, Q)
di, dd, dc = g.dev_malloc(NQ * 80), g.dev_malloc(NQ * 40), g.dev_malloc(NQ * 4)
g.search_batch_dev(dq, NQ, 10, 128, di, dd, dc); g.sync()
t = time.perf_counter()
for _ in range(10): g.search_batch_dev(dq, NQ, 10, 128, di, dd, dc)
g.sync(); dev = (time.perf_counter() - t) / 10
print(f"host buffers (PCIe in/out, pageable): {host*1e3:.2f} ms/batch = {NQ/host:.0f} q/s; HBM resident: {dev*1e3:.2f} ms = {NQ/dev:.0f} q/s; ratio {dev/host:.3f}")
