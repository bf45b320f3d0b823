"""Does the order of the queries inside a batch matter (L2 / MALL reuse between neighbouring wavefronts)?
usage: probe_query_order.py DATASET [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
from bench import gen_vectors
pkg = muninn_amd.pkg
ds = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
D, NQ, K, EF = 768, 10000, 10, 128
X = gen_vectors(N, D, 42, ds); Q = gen_vectors(NQ, D, 43, ds)
g = pkg.HnswIndex(D, "cosine", 16, 200)
t = time.time(); g.build(np.arange(1, N + 1, dtype=np.int64), X, 16, 8192); g.sync(); print(f"build {time.time()-t:.1f}s", flush=True)
d_ids, d_ds, d_cnt = g.dev_malloc(NQ * K * 8), g.dev_malloc(NQ * K * 4), g.dev_malloc(NQ * 4)
def run(Qx, tag):
    dq = g.dev_malloc(Qx.nbytes); g.dev_upload(dq, np.ascontiguousarray(Qx))
    for _ in range(2): g.search_batch_dev(dq, NQ, K, EF, d_ids, d_ds, d_cnt)
    ms = []
    for _ in range(6):
        g.search_batch_dev(dq, NQ, K, EF, d_ids, d_ds, d_cnt); ms.append(g.last_launch()["last_kernel_ms"])
    print(f"{ds} {tag}: kernel {np.mean(ms):.2f} ms  ({NQ/np.mean(ms)*1e3:.0f} q/s)", flush=True)
run(Q, "as generated")
for na in (64, 1024):
    A = X[np.random.default_rng(1).choice(N, na, replace=False)]
    sim = (Q / np.linalg.norm(Q, axis=1, keepdims=True)) @ (A / np.linalg.norm(A, axis=1, keepdims=True)).T
    order = np.argsort(sim.argmax(1), kind="stable")
    run(Q[order], f"grouped by nearest of {na} anchors")
# upper bound of the effect: every query repeated 8 times in a row
run(np.repeat(Q[:NQ // 8], 8, axis=0), "each query 8x in a row")
