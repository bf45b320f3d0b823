"""Single-query latency (the reference's SQL surface answers one query per xFilter) and small-batch search,
one wavefront per query (MN_COOP=0) vs a workgroup per query."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
from bench import gen_vectors
pkg = muninn_amd.pkg
N, D = int(sys.argv[1]), int(sys.argv[2])
X = gen_vectors(N, D, 42, "gaussian"); Q = gen_vectors(64, D, 43, "gaussian")
g = pkg.HnswIndex(D, "cosine", 16, 200)
g.build(np.arange(1, N + 1, dtype=np.int64), X, 16, 8192); g.sync()
ref = None
for coop in ("0", "1"):
    os.environ["MN_COOP"] = coop
    for ef in (64, 128):
        g.search(Q[0], 10, ef)
        t = time.perf_counter(); res = [g.search(q, 10, ef) for q in Q]; dt = (time.perf_counter() - t) / len(Q)
        ids = [tuple(int(x) for x in r[0]) if isinstance(r, tuple) else tuple(int(h[0]) for h in r) for r in res]
        if ref is None: ref = {}
        if ef in ref: assert ref[ef] == ids, "coop result differs"
        ref[ef] = ids
        t = time.perf_counter(); g.search_batch(Q, 10, ef); bt = time.perf_counter() - t
        print(f"{N}x{D} MN_COOP={coop} ef={ef}: single query {dt*1e3:.2f} ms; 64-query batch {bt*1e3:.2f} ms", flush=True)
