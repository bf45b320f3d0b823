"""How often does mn_hnsw_delete refuse (reconnection would exceed a fixed-width row) at realistic parameters?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
rng = np.random.default_rng(3)
for (n, dim, M, efc) in [(20000, 32, 16, 200), (20000, 128, 16, 200), (20000, 32, 8, 100), (5000, 16, 4, 40)]:
    X = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(1, n + 1, dtype=np.int64)
    g = pkg.HnswIndex(dim, "cosine", M, efc)
    g.build(ids, X, 16, 4096)
    ok = ref = 0
    for d in rng.choice(ids, 3000, replace=False):
        r = g.delete(int(d))
        if r == 0: ok += 1
        else: ref += 1
    print(f"n={n} dim={dim} M={M} efc={efc}: {ok} deleted, {ref} refused ({100*ref/(ok+ref):.2f}%)", flush=True)
    g.close()
