"""Schedule pins (tests/golden/schedule_pins.npz, oracle/gen_golden.py schedule_pins): the batch-synchronous HNSW build and
the batched Node2Vec schedule are this repository's own — the reference has neither — so no reference vector pins them.
These digests do: a change of batch sizes, commit order, RNG streams or sample order shows up here on CPU and has to be
made on purpose on both sides (the device is compared with the live oracle by the -m gpu tests, and with the pins below)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import orc
from oracle import orc_graph as og
from oracle.graph_cases import planted

G = os.path.join(os.path.dirname(__file__), "golden")
HNSW = {"l2_m8": (1500, 16, "l2", 8, 60, 5), "cos_m16": (2500, 48, "cosine", 16, 100, 6), "ip_m4": (700, 7, "inner_product", 4, 30, 7)}
N2V = {"b18": ((32, 1.0, 1.0, 4, 30, 4, 4, 0.025, 2), 18), "pq_b7": ((70, 0.5, 2.0, 2, 20, 3, 2, 0.05, 1), 7)}


def _gauss(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def _batches(n, count_fn):
    pos = 0
    while pos < n:  # mn_hnsw_build: grow_div 16, max_batch 8192
        b = max(1, min(count_fn() // 16, 8192, n - pos))
        yield pos, pos + b
        pos += b


@pytest.mark.parametrize("tag", sorted(HNSW))
def test_oracle_batched_build_pin(tag):
    n, d, metric, M, efc, seed = HNSW[tag]
    X, ids = _gauss(n, d, seed), np.arange(1, n + 1, dtype=np.int64)
    o = orc.Oracle(d, metric, M, efc)
    for a, b in _batches(n, lambda: o.node_count):
        assert o.insert_batch(ids[a:b], X[a:b]) == 0
    assert np.array_equal(orc.graph_digest(o.graph(ids)), np.load(os.path.join(G, "schedule_pins.npz"))[f"hnsw_{tag}"])


@pytest.mark.parametrize("tag", sorted(N2V))
def test_oracle_batched_node2vec_pin(tag):
    s, d, _ = planted(1200, 6, 0.08, 0.002, 7)
    prm, batch = N2V[tag]
    emb, npairs = og.node2vec_train_batched(og.N2vGraph(s, d), *prm, batch)
    assert np.array_equal(_sha(emb, np.array([npairs], np.int64)), np.load(os.path.join(G, "schedule_pins.npz"))[f"n2v_{tag}"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(HNSW))
def test_gpu_batched_build_equals_the_pin(gpu, tag):
    n, d, metric, M, efc, seed = HNSW[tag]
    X, ids = _gauss(n, d, seed), np.arange(1, n + 1, dtype=np.int64)
    g = gpu.HnswIndex(d, metric, M, efc)
    assert g.build(ids, X, 16, 8192) == 0
    assert np.array_equal(orc.graph_digest(g.graph(ids)), np.load(os.path.join(G, "schedule_pins.npz"))[f"hnsw_{tag}"])
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(N2V))
def test_gpu_batched_node2vec_equals_the_pin(gpu, tag):
    s, d, _ = planted(1200, 6, 0.08, 0.002, 7)
    gph = og.N2vGraph(s, d)
    (dim, p, q, nw, wl, win, neg, lr, ep), batch = N2V[tag]
    emb, st = gpu.node2vec_train(gph.off, gph.adj, dim, p, q, nw, wl, win, neg, lr, ep, mode=gpu.N2V_BATCHED, batch_walks=batch)
    assert np.array_equal(_sha(emb, np.array([st["pairs"]], np.int64)), np.load(os.path.join(G, "schedule_pins.npz"))[f"n2v_{tag}"])
