"""recall of device batched build vs sequential reference semantics (oracle) at 10k x 128"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
from oracle import orc
pkg = muninn_amd.pkg
N, D = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(42)
X = rng.standard_normal((N, D), dtype=np.float32)
Q = rng.standard_normal((200, D), dtype=np.float32)
ids = np.arange(1, N + 1, dtype=np.int64)
# exact truth by numpy (cosine)
Xn = X / np.linalg.norm(X, axis=1, keepdims=True); Qn = Q / np.linalg.norm(Q, axis=1, keepdims=True)
truth = np.argsort(-(Qn @ Xn.T), axis=1)[:, :10] + 1
def rec(found): return np.mean([len(set(found[i]) & set(truth[i])) / 10 for i in range(len(truth))])
g = pkg.HnswIndex(D, "cosine", 16, 200, order=pkg.ORDER_WAVE)
t = time.time(); g.build(ids, X, 16, 8192); print("gpu batched build", time.time() - t)
dq = g.dev_malloc(Q.nbytes); g.dev_upload(dq, Q)
bf = g.bruteforce_topk(dq, len(Q), 10)
print("bruteforce vs numpy truth recall", rec(bf))
for ef in (20, 64, 128, 256):
    gi, gd, gc = g.search_batch(Q, 10, ef)
    print("gpu-built ef", ef, "recall", rec(gi))
if N <= 20000:
    o = orc.Oracle(D, "cosine", 16, 200)
    t = time.time(); o.insert_many(ids, X); print("oracle sequential build", time.time() - t)
    for ef in (20, 64, 128, 256):
        oi, od, oc = o.search_many(Q, 10, ef)
        print("seq-built ef", ef, "recall", rec(oi))
