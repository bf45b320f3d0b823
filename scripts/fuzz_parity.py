"""Randomised parity sweep on the GPU: device (exact build incl. speculative windows, batched build, search, delete)
against the CPU oracle over random (dim, M, efC, metric, order, n, ef, k) — looks for rare divergences that the
fixed test cases miss.  usage: fuzz_parity.py SECONDS [SEED]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
from oracle import orc
pkg = muninn_amd.pkg
budget = float(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time(); it = 0; bad = 0; refused = 0
while time.time() - t0 < budget:
    it += 1
    dim = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 31, 32, 64, 100, 128, 257]))
    M = int(rng.choice([2, 2, 3, 4, 4, 6, 8, 12, 16, 24, 32, 40, 48, 64, 70, 100]))
    efc = int(rng.choice([1, 5, 10, 40, 100, 200, 300]))
    metric = str(rng.choice(["l2", "cosine", "inner_product"]))
    wave = bool(rng.integers(0, 2))
    n = int(rng.choice([1, 2, 7, 60, 300, 1200, 2500]))
    if M > 32:
        n = min(n, 2 * M + 40)  # exact inserts of wide rows are slow (single wavefront, up to 128 prunes each): just fill the rows
    kind = str(rng.choice(["gauss", "dups", "lattice", "zeros"]))
    if kind == "gauss":
        X = rng.standard_normal((n, dim)).astype(np.float32)
    elif kind == "dups":
        X = rng.standard_normal((max(1, n // 7), dim)).astype(np.float32)[rng.integers(0, max(1, n // 7), n)]
    elif kind == "lattice":
        X = rng.integers(-2, 3, (n, dim)).astype(np.float32)
    else:
        X = rng.standard_normal((n, dim)).astype(np.float32); X[rng.random(n) < 0.2] = 0.0
    ids = rng.permutation(np.arange(1, 10 * n + 1, dtype=np.int64))[:n] - int(rng.integers(0, 5)) * n
    ids = np.unique(ids)[:n]; rng.shuffle(ids); n = len(ids); X = X[:n]
    oo, go = (orc.ORDER_WAVE, pkg.ORDER_WAVE) if wave else (orc.ORDER_SSE, pkg.ORDER_SSE)
    mode = str(rng.choice(["seq", "batched"]))
    tag = f"it={it} dim={dim} M={M} efc={efc} {metric} wave={wave} n={n} {kind} {mode}"
    try:
        o = orc.Oracle(dim, metric, M, efc, order=oo); g = pkg.HnswIndex(dim, metric, M, efc, order=go)
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, 3)]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            if mode == "seq":
                o.insert_many(ids[a:b], X[a:b]); rc = g.insert_batch(ids[a:b], X[a:b], pkg.BUILD_SEQUENTIAL)
            else:
                o.insert_batch(ids[a:b], X[a:b]); rc = g.insert_batch(ids[a:b], X[a:b], pkg.BUILD_BATCHED)
            assert rc == 0, "insert rc"
        assert g.graph(ids) == o.graph(ids), "graph after build"
        nq = int(rng.choice([1, 3, 40, 200])); Q = rng.standard_normal((nq, dim)).astype(np.float32)
        k = int(rng.choice([1, 3, 10, 50])); ef = int(rng.choice([1, 10, 64, 200, 400]))
        wi, wd, wc = o.search_many(Q, k, ef); gi, gd, gc = g.search_batch(Q, k, ef)
        assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.int32), wd.view(np.int32)) and np.array_equal(gc, wc), "search"
        # deletes: a few, or most of the index (reconnection then grows lists past M_max, as in the reference: never refused)
        heavy = bool(rng.integers(0, 4) == 0)
        ndel = min(n, int(n * 0.7) if heavy else int(rng.integers(0, 12)))
        for d in rng.choice(ids, ndel, replace=False):
            ro, rg = o.delete(int(d)), g.delete(int(d))
            if rg == -1 and ro == 0:
                refused += 1
            assert ro == rg, f"delete rc {ro} {rg}: {pkg.hnsw._err()}"
        assert g.graph(ids) == o.graph(ids), "graph after delete"
        gi, gd, gc = g.search_batch(Q, k, max(ef, k)); wi, wd, wc = o.search_many(Q, k, max(ef, k))
        assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.int32), wd.view(np.int32)), "search after delete"
        if n >= 7:  # more inserts after the deletes: lists that outgrew M_max are pruned back by the inserts that touch them
            m2 = int(rng.integers(1, max(2, n // 3)))
            X2 = rng.standard_normal((m2, dim)).astype(np.float32)
            ids2 = np.arange(int(ids.max()) + 1, int(ids.max()) + 1 + m2, dtype=np.int64)
            if mode == "seq":
                o.insert_many(ids2, X2); rc = g.insert_batch(ids2, X2, pkg.BUILD_SEQUENTIAL)
            else:
                o.insert_batch(ids2, X2); rc = g.insert_batch(ids2, X2, pkg.BUILD_BATCHED)
            assert rc == 0, "insert after delete rc"
            allids = np.concatenate([ids, ids2])
            assert g.graph(allids) == o.graph(allids), "graph after re-insert"
        g.close()
    except AssertionError as e:
        bad += 1; print("MISMATCH", tag, "::", e, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1; print("ERROR", tag, "::", repr(e)[:300], flush=True)
    if it % 20 == 0:
        print(f"... {it} cases, {bad} bad, {time.time()-t0:.0f}s", flush=True)
print(f"done: {it} cases, {bad} bad, {refused} delete refusals (must be 0: rows grow as the reference's lists do)")
sys.exit(1 if bad or refused else 0)
