"""lone-query and exact-insert latency at the reference's published sizes; one library per process (MN_AB_LIB=<.so> selects a
variant), answers reduced to a checksum so that two libraries can be compared line by line.
usage: probe_latency3.py [small]"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
if os.environ.get("MN_AB_LIB"):  # another build of the library (same-box comparison of two versions)
    pkg.hnsw.LIB = os.path.abspath(os.environ["MN_AB_LIB"])

def run(n, d, metric, ef, nq=300, nins=300):
    X = np.random.default_rng(42).standard_normal((n + nins * 3, d), dtype=np.float32)
    Q = np.random.default_rng(43).standard_normal((nq, d), dtype=np.float32)
    g = pkg.HnswIndex(d, metric, 16, 200)
    assert g.build(np.arange(1, n + 1, dtype=np.int64), X[:n]) == 0
    out = {}
    crc = 0
    for rep in range(2):
        ts = []
        for i in range(nq):
            t = time.perf_counter(); r = g.search(Q[i], 10, ef); ts.append((time.perf_counter() - t) * 1e3)
            if rep == 0:
                crc = zlib.crc32(r[0].tobytes() + r[1].tobytes(), crc)
        out.setdefault("query_ms", []).append(round(float(np.median(ts[50:])), 4))
    out["answers_crc"] = crc
    pos = n
    t = time.perf_counter()
    for i in range(nins):
        assert g.insert(pos + 1, X[pos]) == 0
        pos += 1
    out["insert_one_at_a_time_per_s"] = round(nins / (time.perf_counter() - t))
    t = time.perf_counter()
    assert g.insert_batch(np.arange(pos + 1, pos + 2 * nins + 1, dtype=np.int64), X[pos:pos + 2 * nins], pkg.BUILD_SEQUENTIAL) == 0
    out["insert_queued_exact_per_s"] = round(2 * nins / (time.perf_counter() - t))
    r = g.search_batch(Q[:64], 10, ef)
    out["graph_crc_after_inserts"] = zlib.crc32(r[0].tobytes() + r[1].tobytes())
    g.close()
    print(f"{n} x {d} {metric} ef={ef}:", out, flush=True)

def floor():
    # an index of one node: the search kernel has nothing to do — what is left is the call itself (ctypes, launch, dispatch, the
    # kernel's prologue and epilogue, the wait)
    g = pkg.HnswIndex(128, "l2", 16, 200)
    X = np.random.default_rng(1).standard_normal((2, 128), dtype=np.float32)
    assert g.insert(1, X[0]) == 0
    ts = []
    for i in range(400):
        t = time.perf_counter(); g.search(X[1], 10, 64); ts.append((time.perf_counter() - t) * 1e3)
    g.close()
    print("one-node index, ms per query:", round(float(np.median(ts[50:])), 4), flush=True)

floor()
run(3000, 128, "l2", 64)
run(10000, 128, "l2", 64)
run(10000, 768, "l2", 64)
if len(sys.argv) < 2:
    run(1000000, 768, "cosine", 128, nq=250, nins=200)
