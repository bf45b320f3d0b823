"""Exact (reference-identical) inserts into an index of BASE vectors: k_insert_seq vs speculative windows.
usage: probe_spec_insert.py BASE DIM N_EXACT   (the base is bulk-built with the batched schedule first)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
BASE, D, NE = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(42)
X = rng.standard_normal((BASE + NE, D), dtype=np.float32)
ids = np.arange(1, BASE + NE + 1, dtype=np.int64)
g = pkg.HnswIndex(D, "cosine", 16, 200)
if BASE:
    t = time.time(); g.build(ids[:BASE], X[:BASE], 16, 8192); g.sync(); print(f"base {BASE}x{D} built in {time.time()-t:.1f}s", flush=True)
t = time.time(); rc = g.insert_batch(ids[BASE:], X[BASE:], pkg.BUILD_SEQUENTIAL); dt = time.time() - t
print(f"MN_SPECULATE={os.environ.get('MN_SPECULATE','1')}: {NE} exact inserts into {BASE}: rc={rc} {dt:.2f}s = {NE/dt:.0f} vec/s", flush=True)
import hashlib
rows = g.export_links(0)
print("links0 sha1", hashlib.sha1(rows[BASE:].tobytes()).hexdigest()[:16], "entry", g.entry_point, g.max_level, flush=True)
