"""Randomised parity sweep of the graph half on the GPU: Leiden (sequential and batched) and Node2Vec (serial and
batched) against the CPU oracle on random small graphs and parameters.  usage: fuzz_graph.py SECONDS [SEED]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
from oracle import orc_graph as og
pkg = muninn_amd.pkg
budget = float(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time(); it = 0; bad = 0
def qb(x): return int(np.float64(x).view(np.int64))
while time.time() - t0 < budget:
    it += 1
    n = int(rng.choice([2, 3, 9, 40, 150, 600, 1500]))
    m = int(max(1, n * float(rng.choice([0.5, 1, 2, 5, 12]))))
    kind = str(rng.choice(["er", "hub", "blocks", "path"]))
    if kind == "er":
        s, d = rng.integers(0, n, m), rng.integers(0, n, m)
    elif kind == "hub":
        s = np.concatenate([np.zeros(min(n - 1, 1400), np.int64), rng.integers(0, n, m)]); d = np.concatenate([np.arange(1, min(n, 1401)), rng.integers(0, n, m)])
    elif kind == "blocks":
        b = rng.integers(0, max(1, n // 20) , n); s = rng.integers(0, n, 4 * m); d = rng.integers(0, n, 4 * m); keep = (b[s] == b[d]) | (rng.random(4 * m) < 0.05); s, d = s[keep], d[keep]
    else:
        s = np.arange(n - 1); d = s + 1
    if len(s) == 0: continue
    if rng.random() < 0.7: keep = s != d; s, d = s[keep], d[keep]   # sometimes keep self loops
    if len(s) == 0: continue
    try:
        if rng.random() < 0.6:   # ---- Leiden ----
            w = None if rng.random() < 0.5 else (rng.integers(1, 9, len(s)) * 0.25 if rng.random() < 0.5 else rng.random(len(s)) * 3 + 0.01)
            res = float(rng.choice([0.3, 1.0, 1.7]))
            csr = og.Csr(s, d, w, "both")
            batch = int(rng.choice([1, 2, 7, 64, 1000, 100000, -2, -3, -3, -4]))  # < 0: whole-graph synchronous sweeps, pick-less period -batch
            tag = f"it={it} leiden n={csr.n} E={len(s)} {kind} weighted={w is not None} res={res} batch={batch}"
            oc, oq, ost = og.leiden(csr, res, batch)
            g = pkg.Graph(csr.n, csr.off_out, csr.tgt_out, csr.w_out if csr.weighted else None, csr.off_in, csr.tgt_in, csr.w_in if csr.weighted else None)
            comm, q, st = g.leiden(res, "both", pkg.LEIDEN_SEQUENTIAL if batch == 1 else pkg.LEIDEN_BATCHED, batch)
            g.close()
            assert np.array_equal(comm, oc), "communities"
            assert qb(q) == qb(oq), f"modularity {q} {oq}"
        else:                    # ---- Node2Vec ----
            dim = int(rng.choice([1, 4, 16, 33, 64, 128, 200])); p = float(rng.choice([1.0, 1.0, 0.5, 2.0])); q_ = float(rng.choice([1.0, 1.0, 0.25, 4.0]))
            nw = int(rng.integers(1, 4)); wl = int(rng.choice([2, 5, 20, 45])); win = int(rng.integers(1, 6)); neg = int(rng.choice([1, 3, 5, 8])); ep = int(rng.integers(1, 3))
            lr = float(rng.choice([0.025, 0.1]))
            gg = og.N2vGraph(s, d)
            serial = gg.n <= 200 and rng.random() < 0.4
            B = int(rng.choice([1, 3, 50, 100000]))
            tag = f"it={it} n2v n={gg.n} E={len(s)} {kind} dim={dim} p={p} q={q_} nw={nw} wl={wl} win={win} neg={neg} ep={ep} serial={serial} B={B}"
            if serial:
                oe, opairs = og.node2vec_train(gg, dim, p, q_, nw, wl, win, neg, lr, ep)
                ge, gst = pkg.node2vec_train(gg.off, gg.adj, dim, p, q_, nw, wl, win, neg, lr, ep, mode=pkg.N2V_SEQUENTIAL)
            else:
                oe, opairs = og.node2vec_train_batched(gg, dim, p, q_, nw, wl, win, neg, lr, ep, B)
                ge, gst = pkg.node2vec_train(gg.off, gg.adj, dim, p, q_, nw, wl, win, neg, lr, ep, mode=pkg.N2V_BATCHED, batch_walks=B)
            assert gst["pairs"] == opairs, f"pairs {gst['pairs']} {opairs}"
            assert np.array_equal(ge.view(np.int32), oe.view(np.int32)), "embedding bits"
    except AssertionError as e:
        bad += 1; print("MISMATCH", tag, "::", e, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1; print("ERROR", tag, "::", repr(e)[:300], flush=True)
    if it % 20 == 0:
        print(f"... {it} cases, {bad} bad, {time.time()-t0:.0f}s", flush=True)
print(f"done: {it} cases, {bad} bad")
