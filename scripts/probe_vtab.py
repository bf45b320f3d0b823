"""Config #1 through the SQL surface: 10k x 128 f32, cosine, M=16 efC=200, one transaction; then 100 kNN queries."""
import os, sqlite3, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = 128
X = np.random.default_rng(42).standard_normal((N + 100, D), dtype=np.float32)
c = sqlite3.connect(":memory:"); c.enable_load_extension(True); c.load_extension(os.path.join(ROOT, "sqlite-muninn_amd", "ext", "muninn"))
c.execute(f"CREATE VIRTUAL TABLE v USING hnsw_index(dimensions={D}, metric='cosine', m=16, ef_construction=200)")
t = time.time()
with c:
    for i in range(N):
        c.execute("INSERT INTO v (rowid, vector) VALUES (?, ?)", (i + 1, X[i].tobytes()))
dt = time.time() - t
print(f"vtab insert: {N} in {dt:.1f}s = {N/dt:.0f} vec/s; edges rows {c.execute('SELECT COUNT(*) FROM v_edges').fetchone()[0]}", flush=True)
Xn = X[:N] / np.linalg.norm(X[:N], axis=1, keepdims=True)
for ef in (64, 128, 256):
    t = time.time(); hit = 0
    for q in X[N:]:
        rows = c.execute("SELECT rowid, distance FROM v WHERE vector MATCH ? AND k = 10 AND ef_search = ?", (q.tobytes(), ef)).fetchall()
        truth = np.argsort(-(Xn @ (q / np.linalg.norm(q))))[:10] + 1
        hit += len(set(r[0] for r in rows) & set(truth.tolist()))
    dt = time.time() - t
    print(f"ef={ef}: {100/dt:.0f} q/s (one xFilter per query, incl. numpy truth), recall@10={hit/1000:.3f}", flush=True)
