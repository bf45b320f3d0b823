"""Seeded graph inputs shared by oracle/gen_golden.py and the tests (TEST INFRASTRUCTURE ONLY).
Edge lists are regenerated from seeds; only expected outputs live in tests/golden/."""
import numpy as np

KARATE = [(1, 2), (1, 3), (1, 4), (1, 5), (1, 6), (1, 7), (1, 8), (1, 9), (1, 11), (1, 12), (1, 13), (1, 14), (1, 18), (1, 20),
          (1, 22), (1, 32), (2, 3), (2, 4), (2, 8), (2, 14), (2, 18), (2, 20), (2, 22), (2, 31), (3, 4), (3, 8), (3, 9), (3, 10),
          (3, 14), (3, 28), (3, 29), (3, 33), (4, 8), (4, 13), (4, 14), (5, 7), (5, 11), (6, 7), (6, 11), (6, 17), (7, 17), (9, 31),
          (9, 33), (9, 34), (10, 34), (14, 34), (15, 33), (15, 34), (16, 33), (16, 34), (19, 33), (19, 34), (20, 34), (21, 33),
          (21, 34), (23, 33), (23, 34), (24, 26), (24, 28), (24, 30), (24, 33), (24, 34), (25, 26), (25, 28), (25, 32), (26, 32),
          (27, 30), (27, 34), (28, 34), (29, 32), (29, 34), (30, 33), (30, 34), (31, 33), (31, 34), (32, 33), (32, 34), (33, 34)]


def er(n, m, seed, weighted=False):
    r = np.random.default_rng(seed)
    s = r.integers(0, n, m)
    d = r.integers(0, n, m)
    keep = s != d
    s, d = s[keep].astype(np.int32), d[keep].astype(np.int32)
    w = (r.random(len(s)) * 3 + 0.1) if weighted else None
    return s, d, w


def planted(n, k, p_in, p_out, seed):
    """k equal blocks; edge inside a block with prob p_in, across with p_out (small n only)."""
    r = np.random.default_rng(seed)
    blk = np.arange(n) % k
    iu, ju = np.triu_indices(n, 1)
    p = np.where(blk[iu] == blk[ju], p_in, p_out)
    keep = r.random(len(iu)) < p
    o = r.permutation(int(keep.sum()))
    return iu[keep][o].astype(np.int32), ju[keep][o].astype(np.int32), None


def leiden_cases():
    """name -> (src, dst, w, resolution); direction is always "both" (the reference's default and the
    only one for which its sweeps are guaranteed to terminate — with out-edges only it can cycle)."""
    ks = np.array(KARATE, np.int32) - 1
    c = {}
    c["barbell"] = (np.array([0, 1, 2, 3, 4, 5, 2], np.int32), np.array([1, 2, 0, 4, 5, 3, 3], np.int32), None, 1.0)
    c["triangle"] = (np.array([0, 1, 2], np.int32), np.array([1, 2, 0], np.int32), None, 1.0)
    c["disconnected"] = (np.array([0, 1, 3, 4], np.int32), np.array([1, 2, 4, 5], np.int32), None, 1.0)
    c["weighted"] = (np.array([0, 1, 2, 2, 3, 4], np.int32), np.array([1, 2, 0, 3, 4, 5], np.int32),
                     np.array([5.0, 5.0, 5.0, 0.1, 5.0, 5.0]), 1.0)
    c["karate"] = (ks[:, 0].copy(), ks[:, 1].copy(), None, 1.0)
    c["karate_r05"] = (ks[:, 0].copy(), ks[:, 1].copy(), None, 0.5)
    c["er200"] = er(200, 600, 1) + (1.0,)
    c["er2000"] = er(2000, 10000, 42) + (1.0,)
    c["er2000w"] = er(2000, 10000, 3, True) + (1.0,)
    c["er500w_r2"] = er(500, 5000, 4, True) + (2.0,)
    c["planted600"] = planted(600, 6, 0.15, 0.005, 7) + (1.0,)
    return c


TWO_CLIQUES = [(1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4), (5, 6), (5, 7), (5, 8), (6, 7), (6, 8), (7, 8), (4, 5)]


def n2v_cases():
    """name -> (edges, (dim, p, q, num_walks, walk_length, window, neg, lr, epochs)); parameters of
    pytests/test_node2vec.py:160,203,233 plus p/q != 1 variants and a random graph with isolated-ish nodes."""
    er_s, er_d, _ = er(60, 150, 11)
    return {
        "cliques16": (TWO_CLIQUES, (16, 1.0, 1.0, 5, 20, 3, 3, 0.025, 3)),
        "cliques32": (TWO_CLIQUES, (32, 1.0, 1.0, 10, 40, 5, 5, 0.025, 5)),
        "cliques_pq": (TWO_CLIQUES, (8, 0.5, 2.0, 4, 15, 2, 2, 0.05, 2)),
        "karate64": (KARATE, (64, 1.0, 1.0, 10, 80, 5, 5, 0.025, 5)),
        "karate_pq": (KARATE, (12, 0.25, 4.0, 3, 30, 4, 3, 0.025, 2)),
        "er60_dim70": ([(int(a), int(b)) for a, b in zip(er_s, er_d)], (70, 2.0, 0.5, 2, 25, 3, 4, 0.03, 1)),
    }


def tvf_cases():
    """name -> (rows of (src, dst) text ids or None, damping or None, iterations or None) for graph_pagerank /
    graph_components: dangling nodes, duplicate rows, self loops, NULLs, isolated pairs, a chain, a star."""
    r = np.random.default_rng(11)
    c = {}
    c["karate"] = ([(str(a), str(b)) for a, b in KARATE], None, None)
    rows = [(f"n{a}", f"n{b}") for a, b in r.integers(0, 40, (120, 2))]
    rows += [("x", "x"), ("y", None), (None, "z"), ("iso1", "iso2"), ("n3", "sink"), ("n3", "sink"), ("sink2", "n5")]
    c["messy"] = (rows, None, None)
    c["messy_d50_i7"] = (rows, 0.5, 7)
    c["chain"] = ([(f"c{i}", f"c{i + 1}") for i in range(60)], 0.85, 30)
    c["star_in"] = ([(f"leaf{i}", "hub") for i in range(50)], None, None)  # 50 sources, one dangling hub
    rows = [(f"v{a}", f"v{b}") for a, b in r.integers(0, 600, (1500, 2))]
    c["sparse600"] = (rows, 0.9, 15)  # many dangling nodes, many components
    c["two_rows"] = ([("a", "b"), ("b", "a")], 0.85, 1)
    c["zero_iterations"] = ([("a", "b"), ("c", "a")], 0.85, 0)
    return c


def betweenness_cases():
    """name -> (rows (src, dst[, w]) as text ids, weighted, direction or None (= "forward"), normalized or None, auto_approx or None)
    for graph_node_betweenness / graph_edge_betweenness: BFS and Dijkstra, the three directions, duplicate rows and self
    loops, equal-length alternatives (sigma > 1), ties between weighted paths, the sqrt(N) source sample."""
    r = np.random.default_rng(21)
    c = {}
    ks = [(str(a), str(b)) for a, b in KARATE]
    c["karate_forward"] = (ks, False, None, None, None)
    c["karate_both_norm"] = (ks, False, "both", 1, None)
    c["karate_reverse"] = (ks, False, "reverse", None, None)
    rows = [(f"n{a}", f"n{b}") for a, b in r.integers(0, 60, (260, 2))] + [("n1", "n1"), ("n2", "n3"), ("n2", "n3")]
    c["er60_forward"] = (rows, False, "forward", None, None)
    c["er60_both"] = (rows, False, "both", None, None)
    c["er60_approx"] = (rows, False, "both", 1, 20)  # 60 nodes > 20 -> ceil(sqrt(60)) = 8 sources, scaled
    wrows = [(f"n{a}", f"n{b}", float(w)) for (a, b), w in zip(r.integers(0, 50, (220, 2)), r.integers(1, 4, 220))]  # small integer weights: many ties
    c["w50_forward"] = (wrows, True, "forward", None, None)
    c["w50_both_norm"] = (wrows, True, "both", 1, None)
    frows = [(f"n{a}", f"n{b}", float(w)) for (a, b), w in zip(r.integers(0, 40, (160, 2)), r.random(160) + 0.1)]
    c["wfloat40_reverse"] = (frows, True, "reverse", None, None)
    grid = [(f"g{i}_{j}", f"g{i + 1}_{j}") for i in range(5) for j in range(6)] + [(f"g{i}_{j}", f"g{i}_{j + 1}") for i in range(6) for j in range(5)]
    c["grid6_both"] = (grid, False, "both", None, None)  # many equal-length shortest paths
    return c
