"""200 lone queries (one k_beam_coop launch each) against a 10k x 128 index — a workload for counter passes.
usage: probe_lone.py [n] [dim]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
if os.environ.get("MN_AB_LIB"):
    pkg.hnsw.LIB = os.path.abspath(os.environ["MN_AB_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
X = np.random.default_rng(42).standard_normal((n, d), dtype=np.float32)
Q = np.random.default_rng(43).standard_normal((200, d), dtype=np.float32)
g = pkg.HnswIndex(d, "l2", 16, 200)
assert g.build(np.arange(1, n + 1, dtype=np.int64), X) == 0
ts = []
for i in range(200):
    t = time.perf_counter(); g.search(Q[i], 10, 64); ts.append((time.perf_counter() - t) * 1e3)
print("median ms", round(float(np.median(ts[20:])), 4), flush=True)
