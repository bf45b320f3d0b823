import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
from oracle import orc_graph as og
from oracle.graph_cases import planted
from importlib import import_module
gr = import_module("sqlite-muninn_amd.graph".replace("-", "_")) if False else pkg.graph
s, d, _ = planted(1200, 6, 0.08, 0.002, 7)
g = og.N2vGraph(s, d)
single, st1 = pkg.node2vec_train(g.off, g.adj, 32, 1.0, 1.0, 2, 20, 3, 3, 0.025, 1, mode=pkg.N2V_BATCHED, batch_walks=50)
L = gr._glib()
c = L.mn_comm_init_host(1, 0, None, None, 0)
prm = gr.N2vParams(32, 1.0, 1.0, 2, 20, 3, 3, 0.025, 1, 50)
out = np.zeros((len(g.off) - 1, 32), np.float32)
st = gr.N2vStats()
rc = L.mn_node2vec_train_shared(c, len(g.off) - 1, np.ascontiguousarray(g.off, np.int32), np.ascontiguousarray(g.adj, np.int32), C.byref(prm), 0, out, C.byref(st))
print("rc", rc, "pairs", st.pairs, st1["pairs"], "equal", np.array_equal(out.view(np.int32), single.view(np.int32)), np.abs(out - single).max())
