"""CPU tests that run only where the compiled reference exists (this build container:
oracle/_ref/libmuninn_ref.so from /root/reference/src).  Randomised live comparison, beyond the
committed fixtures.  Skipped on the GPU box."""
import numpy as np
import pytest

from util import gauss, same_bits


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
