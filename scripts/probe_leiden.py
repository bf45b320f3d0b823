"""config-5 Leiden (LFR-like 500k nodes / 9.3M edges) a few times in one process: device_ms, sweeps, Q.  Meant to be run
bare or under rocprofv3 (--kernel-trace --stats, or --pmc in its own pass).
usage: probe_leiden.py [runs] [nodes] [weighted]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
weighted = len(sys.argv) > 3 and sys.argv[3] == "weighted"
s, d, truth = pkg.lfr.lfr_like(n, 40, min(200, n // 10), 0.3)
w = (np.random.default_rng(8).random(len(s)) * 2 + 0.5) if weighted else None
g = pkg.graph.graph_from_edges(n, s, d, w) if weighted else pkg.graph.graph_from_edges(n, s, d)
for i in range(runs):
    comm, q, st = g.leiden(1.0, "both", pkg.LEIDEN_BATCHED)
    print(f"run {i}: device_ms {st['device_ms']:.2f} sweeps {st['move_sweeps']}+{st['refine_sweeps']} moves {st['moves']} Q {q:.6f} "
          f"comms {comm.max()+1}", flush=True)
g.close()
