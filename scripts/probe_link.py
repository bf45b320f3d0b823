"""Where does the batch link step's time go on concentrated data?  Needs build/ab/linkdbg.so (scripts/build_variant_lib.py linkdbg
-DMN_LINK_DEBUG).  Builds an index over vectors with a strong common component (what node2vec embeddings look like) and prints
k_link_reverse's step counters.  usage: MN_AB_LIB=build/ab/linkdbg.so probe_link.py [n] [common]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import muninn_amd
pkg = muninn_amd.pkg
if os.environ.get("MN_AB_LIB"):  # another build of the library (here: the one with the link step's counters)
    pkg.hnsw.LIB = os.path.join(ROOT, os.environ["MN_AB_LIB"]) if not os.path.isabs(os.environ["MN_AB_LIB"]) else os.environ["MN_AB_LIB"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
common = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
rng = np.random.default_rng(5)
if common < 0:  # node2vec embeddings of an ER graph with config 4's mean degree and parameters (trained by the library under test)
    import bench_graph as bg
    off, adj = pkg.graph.n2v_csr_from_edges(n, *bg.er_edges(n, 20 * n))
    X, _ = pkg.node2vec_train(off, adj, 128, mode=pkg.N2V_BATCHED, p=1.0, q=1.0, num_walks=10, walk_length=80, window=5, neg_samples=5,
                              learning_rate=0.025, epochs=1)
    S = X[:2000] @ X[:2000].T
    print(f"node2vec embeddings: |mean vector| {np.linalg.norm(X.mean(0)):.3f}, cosine similarity of 2000 x 2000: mean {S.mean():.3f} "
          f"max off-diagonal {(S - np.eye(2000)).max():.6f}, exact duplicates of row 0..1999: {int((S > 0.9999999).sum() - 2000)}", flush=True)
X0 = X if common < 0 else None
X = rng.standard_normal((n, 128), dtype=np.float32)
X /= np.linalg.norm(X, axis=1, keepdims=True)
m = rng.standard_normal(128).astype(np.float32); m /= np.linalg.norm(m)
X = X * np.float32(np.sqrt(1 - common ** 2)) + m * np.float32(common)
X /= np.linalg.norm(X, axis=1, keepdims=True)
if X0 is not None:
    X = np.ascontiguousarray(X0, np.float32)
L = pkg.lib()
g = pkg.HnswIndex(128, "cosine", 16, 200)
t = time.time(); assert g.build(np.arange(1, n + 1, dtype=np.int64), X, 16, 8192) == 0; g.sync(); dt = time.time() - t
st = g.build_stats()
print(f"build {n} x 128 (common component {common}): {dt:.2f}s = {n/dt:.0f} vec/s; search {st['search_ms']:.0f} ms, link {st['link_ms']:.0f} ms, "
      f"{st['n_dist']/max(1,st['nodes']):.0f} distances per insert", flush=True)
try:
    L.mn_debug_link_stats
    have = True
except AttributeError:
    have = False
if have:
    out = (C.c_ulonglong * 16)()
    L.mn_debug_link_stats(out, 1)
    names = ["present", "append", "fast_drop", "fast_insert", "prune_ranked", "prune_ties", "targets", "sources", "max_sources_of_a_target",
             "slowest_target_ticks_100MHz", "sum_ticks", "targets_over_1000_sources"]
    print({k: int(out[i]) for i, k in enumerate(names)})
g.close()
