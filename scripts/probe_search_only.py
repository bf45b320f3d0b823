"""search-kernel A/B helper: builds once (WAVE or SSE), times ef=128 10k-query launches"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
N, D = 1000000, 768
order = pkg.ORDER_WAVE if sys.argv[1] == "wave" else pkg.ORDER_SSE
X = np.random.default_rng(42).standard_normal((N, D), dtype=np.float32)
Q = np.random.default_rng(43).standard_normal((10000, D), dtype=np.float32)
g = pkg.HnswIndex(D, "cosine", 16, 200, order=order)
t = time.time(); g.build(np.arange(1, N + 1, dtype=np.int64), X); print("build", time.time() - t, flush=True)
dq = g.dev_malloc(Q.nbytes); g.dev_upload(dq, Q)
di = g.dev_malloc(10000 * 80); dd = g.dev_malloc(10000 * 40); dc = g.dev_malloc(40000)
for ef in (128, 256):
    ks = []
    for r in range(8):
        g.search_batch_dev(dq, 10000, 10, ef, di, dd, dc); st = g.last_launch(); ks.append(st["last_kernel_ms"])
    byt = st["last_n_dist"] * D * 4 + st["last_n_expanded"] * 128 + st["last_n_dist"] * 4
    print(f"{sys.argv[1]} ef={ef}: kernel median {np.median(ks[2:]):.2f} ms min {min(ks):.2f} -> {byt/np.median(ks[2:])/1e6:.0f} GB/s", flush=True)
