"""CPU tests that run only where the compiled reference exists (this build container:
oracle/_ref/libmuninn_ref.so from /root/reference/src).  Randomised live comparison, beyond the
committed fixtures.  Skipped on the GPU box."""
import numpy as np
import pytest

from util import gauss, overgrown_case, same_bits


@pytest.fixture(scope="module")
def ref(orc):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    return orc


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_build_search_delete(ref, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(300, 900))
    d = int(rng.choice([3, 6, 17, 64]))
    M = int(rng.choice([2, 5, 8, 16]))
    efc = int(rng.choice([10, 40, 100]))
    metric = ["l2", "cosine", "inner_product"][seed % 3]
    X = gauss(n, d, seed + 100)
    X[rng.choice(n, n // 20)] = X[0]  # duplicates → distance ties in search and prune
    ids = rng.permutation(np.arange(10_000, 10_000 + n)).astype(np.int64)
    r, o = ref.Ref(d, metric, M, efc, seed=seed), ref.Oracle(d, metric, M, efc, seed=seed)
    half = n // 2
    assert r.insert_many(ids[:half], X[:half]) == 0 and o.insert_many(ids[:half], X[:half]) == 0
    for x in ids[rng.choice(half, half // 8, replace=False)]:
        assert r.delete(int(x)) == o.delete(int(x))
    assert r.insert_many(ids[half:], X[half:]) == 0 and o.insert_many(ids[half:], X[half:]) == 0
    assert r.graph(ids) == o.graph(ids)
    assert r.node_count == o.node_count
    Q = gauss(50, d, seed + 200)
    for k, ef in ((1, 1), (5, 3), (10, 50), (30, 200)):
        a, b = r.search_many(Q, k, ef), o.search_many(Q, k, ef)
        assert np.array_equal(a[0], b[0]) and same_bits(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_lists_grow_past_m_max_and_are_pruned_back_by_later_inserts(ref):
    class R(ref.Ref):  # the reference behind the oracle's load surface
        def load_neighbors(self, id, level, nbrs):
            return self.add_neighbors(id, level, nbrs)

    (r, ids, X, gr), (o, _, _, go) = overgrown_case(R), overgrown_case(ref.Oracle)
    assert gr == go
    assert max(len(v) for (_, l), v in gr["nbrs"].items() if l == 0) > 4  # grown naturally at layer 0 ...
    assert max(len(v) for (_, l), v in gr["nbrs"].items() if l > 0) > 2   # ... and above
    assert r.graph(ids[:1000]) == o.graph(ids[:1000])
    Q = gauss(30, 2, 77)
    a, b = r.search_many(Q, 5, 20), o.search_many(Q, 5, 20)
    assert np.array_equal(a[0], b[0]) and same_bits(a[1], b[1])
    assert r.insert_many(ids[1000:], X[1000:]) == 0 and o.insert_many(ids[1000:], X[1000:]) == 0
    assert r.graph(ids) == o.graph(ids)
    for v in ids[1000::3]:
        assert r.delete(int(v)) == o.delete(int(v)) == 0
    assert r.graph(ids) == o.graph(ids)


def test_node_table_fills_under_churn_as_in_the_reference(ref):
    """The table grows on the LIVE count (:527) while soft-deleted nodes keep their entries: delete + insert churn fills
    the 256-entry table and hnsw_insert fails (ht_insert, :61-74) after consuming its level draw."""
    d = 4
    X = gauss(400, d, 5)
    r, o = ref.Ref(d, "l2", 4, 20), ref.Oracle(d, "l2", 4, 20)
    rc_r, rc_o = [], []
    for i in range(150):
        rc_r.append(r.insert(i + 1, X[i])); rc_o.append(o.insert(i + 1, X[i]))
    for i in range(150):
        rc_r.append(r.delete(i + 1)); rc_o.append(o.delete(i + 1))
    for i in range(150, 400):
        rc_r.append(r.insert(i + 1, X[i])); rc_o.append(o.insert(i + 1, X[i]))
    assert rc_r == rc_o and -1 in rc_r[300:]
    live = [i + 1 for i in range(150, 400) if rc_r[150 + i] == 0]
    assert r.graph(live) == o.graph(live) and r.node_count == o.node_count


def test_large_m_live(ref):
    n, d, M = 500, 6, 80
    X = gauss(n, d, 8)
    ids = np.arange(1, n + 1, dtype=np.int64)
    r, o = ref.Ref(d, "cosine", M, 150), ref.Oracle(d, "cosine", M, 150)
    assert r.insert_many(ids, X) == 0 and o.insert_many(ids, X) == 0
    assert r.graph(ids) == o.graph(ids)


def test_pq_trace_live(ref):
    rng = np.random.default_rng(9)
    ops = (rng.random(400) < 0.6).astype(np.int32)
    d = rng.integers(0, 7, 400).astype(np.float32)
    ids = np.arange(400, dtype=np.int64)
    a, b = ref.ref_pq_trace(ops, ids, d), ref.pq_trace(ops, ids, d)
    assert np.array_equal(a[0], b[0]) and same_bits(a[1], b[1])


def test_committed_sql_transcripts_are_what_the_reference_extension_produces(ref):
    """tests/golden/vtab_fuzz.json.gz and graph_fuzz.json.gz are replayed against OUR extension on the GPU box.  Here,
    where the reference's own extension can be loaded (oracle/_ref/muninn.so), the same generator is run against it
    again: the committed transcripts must be exactly what it produces (no hand editing, generator and fixtures in sync)."""
    import gzip
    import json
    import os
    import sqlite3

    from oracle import gen_golden as gg

    if not os.path.exists(ref.REF_EXT + ".so"):
        pytest.skip("reference extension not built")
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, run, seeds in (("vtab_fuzz.json.gz", gg.vtab_fuzz_run, (0, 5, 11, 100, 105)), ("graph_fuzz.json.gz", gg.graph_fuzz_run, (0, 7))):
        with gzip.open(os.path.join(G, name), "rt") as f:
            want = json.load(f)
        for seed in seeds:
            c = sqlite3.connect(":memory:")
            c.enable_load_extension(True)
            c.load_extension(ref.REF_EXT)
            got = run(c, seed, rollbacks=True) if (name.startswith("vtab") and seed >= 100) else run(c, seed)
            c.close()
            assert got == want[str(seed)], (name, seed)
