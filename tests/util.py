import numpy as np


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))


def gauss(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)


def recall_at_k(found, truth):
    hit = 0
    for f, t in zip(found, truth):
        hit += len(set(int(x) for x in f if x >= 0) & set(int(x) for x in t))
    return hit / float(truth.shape[0] * truth.shape[1])


def overgrown_case(mk, long_lists=True):
    """Index whose lists exceed M_max two ways: naturally (M = 2, 80 % deleted: the reconnection step,
    src/hnsw_algo.c:775-782, grows lists — node_add_neighbor has no bound) and, optionally, through the load API (lists
    of up to 3x M_max, as a database could hold them).  Then more inserts: every insert that touches such a list prunes
    ALL of it back to M_max (:601-646).  Returns the index and the ids for graph()."""
    n, d, M, efc = 1300, 2, 2, 3
    X = gauss(n, d, 31)
    ids = np.arange(1, n + 1, dtype=np.int64)
    x = mk(d, "l2", M, efc)
    assert x.insert_many(ids[:1000], X[:1000]) == 0
    for v in np.random.default_rng(4).permutation(ids[:1000])[:800]:
        assert x.delete(int(v)) == 0
    grown = x.graph(ids[:1000])
    if long_lists:
        rng = np.random.default_rng(6)
        live = [int(i) for i in ids[:1000] if x.node_deleted(int(i)) == 0]
        for i in live[::7]:
            for l in range(x.node_level(i) + 1):
                cand = [c for c in live if c != i and x.node_level(c) >= l]
                extra = rng.choice(cand, min(len(cand), 3 * (2 * M if l == 0 else M)), replace=False)
                assert x.load_neighbors(i, l, np.asarray(extra, np.int64)) == 0
    return x, ids, X, grown
