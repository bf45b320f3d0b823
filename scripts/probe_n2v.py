import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
n = int(sys.argv[1]); m = int(sys.argv[2]); dim = int(sys.argv[3]); nw = int(sys.argv[4]); wl = int(sys.argv[5]); B = int(sys.argv[6]) if len(sys.argv) > 6 else 0
rng = np.random.default_rng(42)
t = time.time()
s = rng.integers(0, n, m); d = rng.integers(0, n, m); keep = s != d; s, d = s[keep], d[keep]
# undirected dedup CSR (node ids = indices; every node assumed present)
a = np.concatenate([s, d]); b = np.concatenate([d, s])
key = a.astype(np.int64) * n + b
_, first = np.unique(key, return_index=True)
first.sort()
a, b = a[first], b[first]
o = np.argsort(a, kind="stable")
adj = b[o].astype(np.int32)
off = np.zeros(n + 1, np.int64); np.add.at(off, a + 1, 1); off = np.cumsum(off).astype(np.int32)
print(f"graph n={n} directed edges={len(adj)} gen {time.time()-t:.1f}s", flush=True)
t = time.time()
emb, st = pkg.node2vec_train(off, adj, dim, 1.0, 1.0, nw, wl, 5, 5, 0.025, 1, mode=pkg.N2V_BATCHED, batch_walks=B)
dt = time.time() - t
pairs = st["pairs"]
byt = pairs * (2 * 6 + 2) * dim * 4
print(f"batched: {dt:.2f}s wall, device {st['device_ms']/1e3:.2f}s, pairs {pairs} -> {pairs/dt/1e6:.1f} M pairs/s, alg {byt/st['device_ms']/1e6:.0f} GB/s", flush=True)
