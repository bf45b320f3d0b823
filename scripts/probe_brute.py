import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
rng = np.random.default_rng(1)
for metric in ("inner_product", "l2", "cosine"):
    for (n, dim, nq, k) in [(300, 8, 4, 5), (5000, 50, 300, 10)]:
        X = rng.standard_normal((n, dim)).astype(np.float32)
        Q = rng.standard_normal((nq, dim)).astype(np.float32)
        ids = np.arange(n, dtype=np.int64)
        g = pkg.HnswIndex(dim, metric, 8, 40)
        g.insert_batch(ids, X, pkg.BUILD_BATCHED)
        dq = g.dev_malloc(Q.nbytes); g.dev_upload(dq, Q)
        got = g.bruteforce_topk(dq, nq, k)
        os.environ["MN_BRUTE"] = "valu"
        valu = g.bruteforce_topk(dq, nq, k)
        del os.environ["MN_BRUTE"]
        D = {"inner_product": -(Q.astype(np.float64) @ X.T.astype(np.float64)),
             "l2": ((Q[:, None, :].astype(np.float64) - X[None].astype(np.float64)) ** 2).sum(2),
             "cosine": 1 - (Q @ X.T) / (np.linalg.norm(Q, axis=1)[:, None] * np.linalg.norm(X, axis=1)[None])}[metric]
        want = np.argsort(D, axis=1, kind="stable")[:, :k]
        print(metric, n, dim, nq, k, "got==want", np.mean(got == want), "valu==want", np.mean(valu == want))
        if np.mean(got == want) < 0.9:
            print(" got ", got[0], [round(float(D[0, i]), 3) if i >= 0 else None for i in got[0]])
            print(" want", want[0], [round(float(D[0, i]), 3) for i in want[0]])
        g.close()
