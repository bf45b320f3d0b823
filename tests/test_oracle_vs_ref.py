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
