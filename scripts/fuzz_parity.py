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
    M = int(rng.choice([2, 3, 4, 6, 8, 12, 16, 24, 32, 40, 48, 64]))
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
        for d in rng.choice(ids, min(n, int(rng.integers(0, 12))), replace=False):
            ro, rg = o.delete(int(d)), g.delete(int(d))
            if rg == -1 and ro == 0:
                msg = pkg.hnsw._err() or ""
                refused += 1
                assert "row" in msg or "overflow" in msg or "width" in msg, f"delete refused with: {msg}"
                break  # documented refusal: reconnection would overflow a fixed-width row; index untouched → stop deleting
            assert ro == rg, f"delete rc {ro} {rg}"
        else:
            assert g.graph(ids) == o.graph(ids), "graph after delete"
            gi, gd, gc = g.search_batch(Q, k, max(ef, k)); wi, wd, wc = o.search_many(Q, k, max(ef, k))
            assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.int32), wd.view(np.int32)), "search after delete"
        g.close()
    except AssertionError as e:
        bad += 1; print("MISMATCH", tag, "::", e, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1; print("ERROR", tag, "::", repr(e)[:300], flush=True)
    if it % 20 == 0:
        print(f"... {it} cases, {bad} bad, {time.time()-t0:.0f}s", flush=True)
print(f"done: {it} cases, {bad} bad, {refused} delete refusals (fixed-width rows)")
