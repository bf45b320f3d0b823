"""A/B of the batched Leiden schedule's tuning knobs on the config-5 graph (one graph, several settings).
usage: leiden_ab.py [GROW settings like 1,1 8,256 4,256]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
if os.environ.get("MN_AB_LIB"):  # another build of the library (same-box comparison of two versions)
    pkg.hnsw.LIB = os.environ["MN_AB_LIB"]
n = 500_000
s, d, truth = pkg.lfr.lfr_like(n, 40, 200, 0.3)
g = pkg.graph.graph_from_edges(n, s, d)
for cfg in (sys.argv[1:] or ["8,256"]):
    os.environ["MN_LEIDEN_GROW"] = cfg
    g.leiden(1.0, "both", pkg.LEIDEN_BATCHED)
    ms = []
    for _ in range(3):
        comm, q, st = g.leiden(1.0, "both", pkg.LEIDEN_BATCHED)
        ms.append(st["device_ms"])
    print(f"GROW={cfg}: device_ms {min(ms):.2f} (mean {np.mean(ms):.2f}) sweeps {st['move_sweeps']}+{st['refine_sweeps']} Q {q:.6f} comms {comm.max()+1}", flush=True)
g.close()
