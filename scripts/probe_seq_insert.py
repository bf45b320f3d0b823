import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
N, D = int(sys.argv[1]), int(sys.argv[2])
X = np.random.default_rng(42).standard_normal((N, D), dtype=np.float32)
g = pkg.HnswIndex(D, "cosine", 16, 200)
t = time.time(); g.insert_batch(np.arange(1, N + 1, dtype=np.int64), X, pkg.BUILD_SEQUENTIAL); dt = time.time() - t
st = g.last_launch()
print(f"sequential (exact) build {N}x{D}: {dt:.2f}s = {N/dt:.0f} vec/s; kernel {st['last_kernel_ms']:.0f} ms; n_dist/insert {st['last_n_dist']/N:.0f}")
