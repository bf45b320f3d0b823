import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
N, D = int(sys.argv[1]), int(sys.argv[2])
order = pkg.ORDER_WAVE if (len(sys.argv) > 3 and sys.argv[3] == "wave") else pkg.ORDER_SSE
X = np.random.default_rng(42).standard_normal((N + 200, D), dtype=np.float32)
g = pkg.HnswIndex(D, "cosine", 16, 200, order=order)
g.build(np.arange(1, N + 1, dtype=np.int64), X[:N])
for ef in (64, 128):
    for nq in (1, 8, 64):
        ks, ws = [], []
        for r in range(30):
            Q = X[N + r:N + r + nq] if nq == 1 else X[N:N + nq]
            t = time.perf_counter(); g.search_batch(Q, 10, ef); ws.append(time.perf_counter() - t)
            st = g.last_launch(); ks.append(st["last_kernel_ms"])
        print(f"ef={ef} nq={nq}: wall {np.median(ws)*1e3:.3f} ms, kernel {np.median(ks):.3f} ms, n_dist/q {st['last_n_dist']/nq:.0f} exp/q {st['last_n_expanded']/nq:.0f}", flush=True)
t = time.perf_counter()
for i in range(200):
    g.insert(N + 1 + i, X[N + i] if i < 200 else X[i])
print(f"single insert: {(time.perf_counter()-t)/200*1e3:.2f} ms each", flush=True)
