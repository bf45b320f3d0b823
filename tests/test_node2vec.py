"""Node2Vec (src/node2vec.c): the oracle's serial restatement against the embedding bytes the compiled
reference produced through its own SQL function (tests/golden/node2vec.npz), and the HIP sequential
kernel against the same bytes.  Plus the reference's statistical acceptance tests
(pytests/test_node2vec.py:194-273)."""
import os

import numpy as np
import pytest

from oracle import orc_graph as og
from oracle.graph_cases import n2v_cases

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = n2v_cases()


def _graph(edges):
    e = np.array(edges, np.int32)
    return og.N2vGraph(e[:, 0], e[:, 1])


def _within_between(emb, index_of_id, a_ids, b_ids):
    def cos(x, y):
        return float(np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y)))

    A = [emb[index_of_id[i]] for i in a_ids]
    B = [emb[index_of_id[i]] for i in b_ids]
    within = [cos(A[i], A[j]) for i in range(len(A)) for j in range(i + 1, len(A))]
    within += [cos(B[i], B[j]) for i in range(len(B)) for j in range(i + 1, len(B))]
    between = [cos(x, y) for x in A for y in B]
    return np.mean(within), np.mean(between)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_bytes_equal_reference(name):
    z = np.load(os.path.join(G, "node2vec.npz"))
    edges, prm = CASES[name]
    emb, npairs = og.node2vec_train(_graph(edges), *prm)
    assert np.array_equal(emb.view(np.int32), z[name]), name


def test_first_seen_dedup_graph():
    """graph_load_edges (src/node2vec.c:112-138): first-seen indices, both directions, duplicates dropped"""
    g = og.N2vGraph(np.array([5, 5, 7, 2], np.int32), np.array([7, 7, 5, 5], np.int32))
    assert g.n == 3 and g.index_of_id[5] == 0 and g.index_of_id[7] == 1 and g.index_of_id[2] == 2
    assert g.off.tolist() == [0, 2, 3, 4] and g.adj.tolist() == [1, 2, 0, 0]


def test_acceptance_karate_within_gt_between():
    edges, prm = CASES["karate64"]
    g = _graph(edges)
    emb, _ = og.node2vec_train(g, *prm)
    a = [1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 14, 17, 18, 20, 22]
    b = [9, 10, 15, 16, 19, 21, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34]
    w, bt = _within_between(emb, g.index_of_id, a, b)
    assert w > bt


def test_walk_properties():
    edges, _ = CASES["karate64"]
    g = _graph(edges)
    walk, st = og.biased_walk(g, 0, 0.5, 2.0, 40, 12345)
    assert len(walk) == 40 and walk[0] == 0
    for a, b in zip(walk[:-1], walk[1:]):
        assert b in g.adj[g.off[a]:g.off[a + 1]]
    walk2, st2 = og.biased_walk(g, 0, 0.5, 2.0, 40, 12345)
    assert np.array_equal(walk, walk2) and st == st2


# ───────────────────────── GPU ─────────────────────────

@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_sequential_bytes_equal_reference(gpu, name):
    z = np.load(os.path.join(G, "node2vec.npz"))
    edges, prm = CASES[name]
    g = _graph(edges)
    dim, p, q, nw, wl, win, neg, lr, ep = prm
    emb, st = gpu.node2vec_train(g.off, g.adj, dim, p, q, nw, wl, win, neg, lr, ep)
    assert np.array_equal(emb.view(np.int32), z[name]), name
    _, npairs = og.node2vec_train(g, *prm)
    assert st["pairs"] == npairs


@pytest.mark.gpu
def test_gpu_acceptance_two_cliques(gpu):
    edges, prm = CASES["cliques32"]
    g = _graph(edges)
    dim, p, q, nw, wl, win, neg, lr, ep = prm
    emb, _ = gpu.node2vec_train(g.off, g.adj, dim, p, q, nw, wl, win, neg, lr, ep)
    w, b = _within_between(emb, g.index_of_id, [1, 2, 3, 4], [5, 6, 7, 8])
    assert w > b
    assert np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_gpu_empty_and_invalid(gpu):
    emb, st = gpu.node2vec_train(np.zeros(1, np.int32), np.zeros(0, np.int32), 8)
    assert emb.shape == (0, 8)
    with pytest.raises(Exception):
        gpu.node2vec_train(np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32), 2000)


def _planted_graph():
    from oracle.graph_cases import planted

    s, d, _ = planted(1200, 6, 0.08, 0.002, 7)
    return og.N2vGraph(s, d)


def _block_quality(g, emb):
    blk = np.arange(1200) % 6
    E = emb[g.index_of_id[np.arange(1200)]]
    S = E @ E.T
    same = blk[:, None] == blk[None, :]
    np.fill_diagonal(same, False)
    return float(S[same].mean()), float(S[~same].mean())


def test_batched_schedule_quality_matches_serial_on_cpu():
    g = _planted_graph()
    prm = (32, 1.0, 1.0, 4, 30, 4, 4, 0.025, 2)
    ws, bs = _block_quality(g, og.node2vec_train(g, *prm)[0])
    wb, bb = _block_quality(g, og.node2vec_train_batched(g, *prm, 18)[0])
    assert ws - bs > 0.4 and wb - bb > 0.4 and abs((wb - bb) - (ws - bs)) < 0.15


@pytest.mark.gpu
@pytest.mark.parametrize("prm,batch", [((32, 1.0, 1.0, 4, 30, 4, 4, 0.025, 2), 18), ((32, 1.0, 1.0, 2, 20, 3, 3, 0.025, 1), 1200),
                                       ((70, 0.5, 2.0, 2, 20, 3, 2, 0.05, 1), 7), ((130, 2.0, 0.5, 1, 12, 2, 3, 0.02, 1), 64)])
def test_gpu_batched_bit_exact_vs_oracle_schedule(gpu, prm, batch):
    g = _planted_graph()
    dim, p, q, nw, wl, win, neg, lr, ep = prm
    want, npairs = og.node2vec_train_batched(g, *prm, batch)
    got, st = gpu.node2vec_train(g.off, g.adj, dim, p, q, nw, wl, win, neg, lr, ep, mode=gpu.N2V_BATCHED, batch_walks=batch)
    assert st["pairs"] == npairs
    assert np.array_equal(got.view(np.int32), want.view(np.int32))


@pytest.mark.gpu
def test_gpu_batched_acceptance_blocks(gpu):
    g = _planted_graph()
    emb, st = gpu.node2vec_train(g.off, g.adj, 32, 1.0, 1.0, 4, 30, 4, 4, 0.025, 2, mode=gpu.N2V_BATCHED)
    w, b = _block_quality(g, emb)
    assert w - b > 0.4  # within-block similarity far above between-block (pytests/test_node2vec.py:194-273 in spirit)


@pytest.mark.gpu
def test_gpu_batched_bit_exact_vs_oracle_schedule_8k_nodes(gpu):
    """config 4's kernels at a size between the unit graphs and the bench: ER graph, 8 000 nodes / 80 000 edge draws,
    dim 32, batches of 1024 walks — embedding bits and pair count equal the CPU restatement of the same schedule."""
    rng = np.random.default_rng(3)
    s, d = rng.integers(0, 8000, 80_000), rng.integers(0, 8000, 80_000)
    keep = s != d
    g = og.N2vGraph(s[keep], d[keep])
    prm = (32, 1.0, 1.0, 2, 20, 5, 5, 0.025, 1)
    want, npairs = og.node2vec_train_batched(g, *prm, 1024)
    got, st = gpu.node2vec_train(g.off, g.adj, 32, 1.0, 1.0, 2, 20, 5, 5, 0.025, 1, mode=gpu.N2V_BATCHED, batch_walks=1024)
    assert st["pairs"] == npairs and np.array_equal(got.view(np.int32), want.view(np.int32))


@pytest.mark.gpu
def test_config4_shape_properties_200k_nodes(gpu):
    """BASELINE config 4's parameters (p = q = 1, dim 128, window 5, neg 5, 80-step walks) on a 200k-node / 4M-edge-draw ER
    graph: every embedding is unit length (the reference normalises before its INSERTs), the pair count is the closed
    form for walks that never dead-end, and a second run gives the same bits.  (An ER graph has no community structure for
    the embeddings to separate: the quality checks live in the planted-block tests above.)"""
    rng = np.random.default_rng(42)
    n, m = 200_000, 4_000_000
    s, d = rng.integers(0, n, m), rng.integers(0, n, m)
    keep = s != d
    off, adj = gpu.graph.n2v_csr_from_edges(n, s[keep], d[keep])
    prm = dict(p=1.0, q=1.0, num_walks=2, walk_length=80, window=5, neg_samples=5, learning_rate=0.025, epochs=1)
    emb, st = gpu.node2vec_train(off, adj, 128, mode=gpu.N2V_BATCHED, **prm)
    emb2, st2 = gpu.node2vec_train(off, adj, 128, mode=gpu.N2V_BATCHED, **prm)
    assert np.array_equal(emb.view(np.int32), emb2.view(np.int32)) and st["pairs"] == st2["pairs"]
    nn = len(off) - 1
    assert np.abs(np.linalg.norm(emb, axis=1) - 1.0).max() < 1e-5
    L, W = 80, 5  # pairs of one full-length walk: sum over positions of the clipped window (src/node2vec.c:519-531)
    per_walk = sum(min(L - 1, i + W) - max(0, i - W) for i in range(L))
    assert st["pairs"] == nn * 2 * per_walk  # (mean degree 40: no isolated nodes, no dead ends on an undirected graph)
    assert np.isfinite(emb).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [64, 30])
def test_train_into_index_keeps_the_embeddings_in_hbm_and_builds_the_same_graph(gpu, dim):
    """mn_node2vec_train_into (config 4's "-> hnsw0 index" leg, src/node2vec.c:540-583): the embeddings are trained, normalised
    and handed to the index build inside HBM.  Same embedding bytes as mn_node2vec_train in the same mode, and the index is the
    graph mn_hnsw_build makes from a host copy of those embeddings with the reference's rowids (first-seen index + 1).
    (dim 30: the index pads rows to 32 floats, so the device-to-device hand-off is a strided copy.)"""
    from oracle.graph_cases import planted

    s, d, _ = planted(3000, 6, 0.05, 0.001, 11)
    g = og.N2vGraph(s, d)
    want, st1 = gpu.node2vec_train(g.off, g.adj, dim, 1.0, 1.0, 3, 20, 4, 3, 0.025, 1, mode=gpu.N2V_BATCHED, batch_walks=200)
    ix = gpu.HnswIndex(dim, "cosine", 8, 60)
    emb, st = gpu.graph.node2vec_train_into(g.off, g.adj, dim, ix, 1, True, 1.0, 1.0, 3, 20, 4, 3, 0.025, 1, 200)
    assert np.array_equal(emb.view(np.int32), want.view(np.int32)) and st["pairs"] == st1["pairs"]
    ids = np.arange(1, g.n + 1, dtype=np.int64)
    ref = gpu.HnswIndex(dim, "cosine", 8, 60)
    assert ref.build(ids, want, 16, 8192) == 0
    assert ix.node_count == g.n and ix.entry_point == ref.entry_point and ix.max_level == ref.max_level
    assert np.array_equal(ix.export_links(0), ref.export_links(0)) and np.array_equal(ix.export_links(1), ref.export_links(1))
    q = want[:50]
    a, b = ix.search_batch(q, 5, 40), ref.search_batch(q, 5, 40)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.int32), b[1].view(np.int32))
    ix.close()
    ref.close()
