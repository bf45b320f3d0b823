"""lone-query and exact-insert latency at several sizes in ONE process, with the speculative row requests on and off
(MN_SPEC_ROWS is read per call, so both settings run against the same index on the same box).
usage: probe_latency2.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg

def run(n, d, metric, ef, nq=300, nins=300):
    X = np.random.default_rng(42).standard_normal((n + nins * 2, d), dtype=np.float32)
    Q = np.random.default_rng(43).standard_normal((nq, d), dtype=np.float32)
    g = pkg.HnswIndex(d, metric, 16, 200)
    assert g.build(np.arange(1, n + 1, dtype=np.int64), X[:n]) == 0
    out = {}
    ref = None
    for spec in ("1", "0", "1", "0"):
        os.environ["MN_SPEC_ROWS"] = spec
        ts, ids = [], []
        for i in range(nq):
            t = time.perf_counter(); r = g.search(Q[i], 10, ef); ts.append((time.perf_counter() - t) * 1e3); ids.append(r)
        if ref is None:
            ref = ids
        same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.int32), b[1].view(np.int32)) for a, b in zip(ref, ids))
        out.setdefault("query_ms_spec" + spec, []).append(round(float(np.median(ts[50:])), 4))
        out["same_answers"] = out.get("same_answers", True) and same
    pos = n
    for spec in ("1", "0"):
        os.environ["MN_SPEC_ROWS"] = spec
        t = time.perf_counter()
        for i in range(nins):
            assert g.insert(pos + 1, X[pos]) == 0
            pos += 1
        out["insert_one_at_a_time_per_s_spec" + spec] = round(nins / (time.perf_counter() - t))
    g.close()
    print(f"{n} x {d} {metric} ef={ef}:", out, flush=True)

run(3000, 128, "l2", 64)
run(10000, 128, "l2", 64)
run(10000, 768, "l2", 64)
run(1000000, 768, "cosine", 128, nq=250, nins=200)
