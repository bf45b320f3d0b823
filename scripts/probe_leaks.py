"""Device-memory leak check: create / use / destroy indexes, graphs, node2vec sessions and SQL tables in a loop."""
import os, sys, sqlite3
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd, torch
pkg = muninn_amd.pkg
def free_mb(): torch.cuda.synchronize(); return torch.cuda.mem_get_info(0)[0] / 2**20
rng = np.random.default_rng(1)
X = rng.standard_normal((3000, 32)).astype(np.float32); ids = np.arange(1, 3001, dtype=np.int64)
s, d = rng.integers(0, 2000, 8000), rng.integers(0, 2000, 8000)
off, adj = pkg.graph.n2v_csr_from_edges(2000, s, d)
ext = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sqlite-muninn_amd", "ext", "muninn")
def once():
    g = pkg.HnswIndex(32, "cosine", 8, 60); g.build(ids, X, 16, 512); g.insert_batch(ids[:50] + 10000, X[:50], pkg.BUILD_SEQUENTIAL)
    g.search_batch(X[:200], 5, 40); g.search(X[0], 5, 40); g.delete(5); g.take_dirty(); g.edges_of(ids[:10]); g.close()
    gr = pkg.graph.graph_from_edges(2000, s, d); gr.leiden(1.0, "both", pkg.LEIDEN_BATCHED); gr.leiden(1.0, "both", pkg.LEIDEN_SEQUENTIAL); gr.close()
    pkg.node2vec_train(off, adj, 16, 1.0, 1.0, 1, 10, 3, 3, 0.025, 1, mode=pkg.N2V_BATCHED)
    c = sqlite3.connect(":memory:"); c.enable_load_extension(True); c.load_extension(ext)
    c.execute("CREATE VIRTUAL TABLE t USING hnsw_index(dimensions=32, metric='l2', m=4)")
    with c: c.executemany("INSERT INTO t (rowid, vector) VALUES (?,?)", [(int(i), X[i].tobytes()) for i in range(200)])
    c.execute("SELECT rowid FROM t WHERE vector MATCH ? AND k=3", (X[0].tobytes(),)).fetchall(); c.execute("DROP TABLE t"); c.close()
once(); once()
a = free_mb()
for i in range(40): once()
b = free_mb()
print(f"free before {a:.1f} MiB, after 40 rounds {b:.1f} MiB, delta {a-b:.1f} MiB")
assert a - b < 64, "device memory leak"
