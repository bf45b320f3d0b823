"""exact / deferred mode INSERTs through SQL with MUNINN_PROFILE=1: device time vs shadow-table time.  usage: probe_sql_exact.py [rows] [dim] [mode]"""
import os, sys, sqlite3, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
os.environ["MUNINN_HNSW_MODE"] = sys.argv[3] if len(sys.argv) > 3 else "exact"
os.environ["MUNINN_PROFILE"] = "1"
X = np.random.default_rng(42).standard_normal((n, dim), dtype=np.float32)
c = sqlite3.connect(":memory:")
c.enable_load_extension(True)
c.load_extension(os.path.join(ROOT, "sqlite-muninn_amd", "ext", "muninn"))
c.execute(f"CREATE VIRTUAL TABLE t USING hnsw_index(dimensions={dim}, metric='l2', m=16, ef_construction=200)")
t0 = time.perf_counter()
for i in range(n):
    c.execute("INSERT INTO t (rowid, vector) VALUES (?, ?)", (i + 1, X[i].tobytes()))
c.execute("SELECT rowid FROM t WHERE rowid = 1").fetchall()
dt = time.perf_counter() - t0
print(f"{os.environ['MUNINN_HNSW_MODE']}: {n} rows x {dim}: {n / dt:.0f} rows/s ({dt / n * 1e3:.2f} ms per row)", flush=True)
c.commit()
c.close()
