"""Leiden batched mode: time / modularity / sweeps as a function of the round size (cfg5 graph)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
n = int(sys.argv[1]); batches = [int(x) for x in sys.argv[2].split(",")]
s, d, truth = pkg.lfr.lfr_like(n, 40, min(200, n // 10), 0.3)
g = pkg.graph.graph_from_edges(n, s, d)
g.leiden(1.0, "both", pkg.LEIDEN_BATCHED)
for b in batches:
    t = time.time(); comm, q, st = g.leiden(1.0, "both", pkg.LEIDEN_BATCHED, b); dt = time.time() - t
    print(f"batch={b}: {dt*1e3:.1f} ms Q={q:.5f} K={comm.max()+1} sweeps={st['move_sweeps']}+{st['refine_sweeps']} moves={st['moves']} dev={st['device_ms']:.1f}", flush=True)
