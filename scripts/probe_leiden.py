import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
from oracle import orc_graph as og
pkg = muninn_amd.pkg
n = int(sys.argv[1]); deg = int(sys.argv[2]); modes = sys.argv[3].split(","); batch = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
from importlib import import_module
lfr = pkg.lfr if hasattr(pkg, "lfr") else import_module("sqlite_muninn_amd.lfr")
t = time.time(); s, d, truth = lfr.lfr_like(n, deg, min(200, n // 10), 0.3); print(f"gen n={n} E={len(s)} {time.time()-t:.1f}s", flush=True)
t = time.time(); csr = og.Csr(s, d, None, "both", n_nodes=n, first_seen=False); print(f"csr {time.time()-t:.1f}s", flush=True)
g = pkg.Graph(csr.n, csr.off_out, csr.tgt_out, None, csr.off_in, csr.tgt_in, None)
for mode in modes:
    t = time.time()
    if mode == "cpu":
        comm, q, st = og.leiden(csr, 1.0, 1)
    elif mode == "cpub":
        comm, q, st = og.leiden(csr, 1.0, batch)
    elif mode == "seq":
        comm, q, st = g.leiden(1.0, "both", pkg.LEIDEN_SEQUENTIAL)
    else:
        comm, q, st = g.leiden(1.0, "both", pkg.LEIDEN_BATCHED, batch)
    print(f"{mode}: {time.time()-t:.2f}s Q={q:.5f} K={comm.max()+1} {st}", flush=True)
