"""Where the latency chain of ONE search goes (per expansion: heap pop / link row + visited probe / distances / heap pushes),
from the -DMN_PHASE_TIMING build (scripts/build_phase_lib.py).  usage: probe_phases.py [N] [dim] [metric]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import muninn_amd
pkg = muninn_amd.pkg
pkg.hnsw.LIB = os.path.join(ROOT, "scripts", "_phase", "libmuninn_hip.so")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
L = pkg.hnsw.lib()
for f in ("mn_debug_phase_kernels", "mn_debug_phase_seq"):
    getattr(L, f).argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
def phases(which, reset=True):
    a = (C.c_ulonglong * 8)()
    assert getattr(L, "mn_debug_phase_" + which)(a, 1 if reset else 0) == 0
    return np.array(list(a), np.float64)
def show(tag, p, wall_ms, n):
    us = p[:4] / 100.0  # 100 MHz ticks -> microseconds
    nexp = max(p[5], 1)
    print(f"{tag}: {wall_ms / n:.3f} ms each; expansions {p[5] / n:.0f}, pushes {p[4] / n:.0f}; per expansion: pop {us[0] / nexp:.2f} us, "
          f"row+visited {us[1] / nexp:.2f}, distances {us[2] / nexp:.2f}, pushes {us[3] / nexp:.2f}  (sum {us.sum() / nexp:.2f} us; "
          f"in-search total {us.sum() / n / 1e3:.3f} ms; link phase {p[6] / 100.0 / n / 1e3:.3f} ms; pops of the previous runner-up {p[7] / nexp:.2f})", flush=True)
X = np.random.default_rng(42).standard_normal((N + 400, D), dtype=np.float32)
Q = np.random.default_rng(43).standard_normal((200, D), dtype=np.float32)
g = pkg.HnswIndex(D, metric, 16, 200)
t = time.perf_counter(); g.build(np.arange(1, N + 1, dtype=np.int64), X[:N]); print(f"built {N}x{D} in {time.perf_counter() - t:.1f}s", flush=True)
for ef in (64, 128):
    for i in range(20): g.search(Q[i], 10, ef)
    phases("kernels")
    t = time.perf_counter()
    for i in range(200): g.search(Q[i], 10, ef)
    show(f"single search ef={ef}", phases("kernels"), (time.perf_counter() - t) * 1e3, 200)
os.environ["MN_SPECULATE"] = "0"
for i in range(10): g.insert(N + 1 + i, X[N + i])
phases("seq")
t = time.perf_counter()
for i in range(10, 210): g.insert(N + 1 + i, X[N + i])
show("single insert (k_insert_seq, efC=200)", phases("seq"), (time.perf_counter() - t) * 1e3, 200)
