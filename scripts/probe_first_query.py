"""what the FIRST lone query of a process costs after exact inserts only (the kernels of its translation unit have not run yet):
MN_LAZY_MODULES=1 = code objects loaded at first use (HIP's default), default = loaded when the index is created.
usage: probe_first_query.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
X = np.random.default_rng(42).standard_normal((400, 128), dtype=np.float32)
t = time.perf_counter(); g = pkg.HnswIndex(128, "l2", 16, 200); t_create = time.perf_counter() - t
t = time.perf_counter()
for i in range(300):
    assert g.insert(i + 1, X[i]) == 0
t_ins = time.perf_counter() - t
ts = []
for i in range(300, 320):
    t = time.perf_counter(); g.search(X[i], 10, 64); ts.append((time.perf_counter() - t) * 1e3)
print(f"lazy={os.environ.get('MN_LAZY_MODULES', '0')}: create {t_create * 1e3:.1f} ms, 300 exact inserts {t_ins * 1e3:.0f} ms, first queries (ms):",
      [round(x, 3) for x in ts[:4]], "median of the rest", round(float(np.median(ts[4:])), 3), flush=True)
