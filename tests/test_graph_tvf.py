"""graph_pagerank / graph_components (src/graph_tvf.c): the oracle against what the reference's own SQL
functions returned (CPU), the HIP kernels against the oracle and those goldens (GPU, rank bits / root ids exact), and
the two table-valued functions of the extension against the same goldens through SQL."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import orc_graph as og
from oracle.graph_cases import tvf_cases

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = tvf_cases()


def golden():
    with gzip.open(os.path.join(G, "graph_tvf.json.gz"), "rt") as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_golden(name):
    rows, damping, iterations = CASES[name]
    z = golden()[name]
    ids, s, d = og.first_seen_edges(rows)
    assert ids == z["nodes"] == z["comp_nodes"]  # first-seen order: src of a row before its dst, NULL rows skipped
    pr = og.pagerank(len(ids), s, d, 0.85 if damping is None else damping, 20 if iterations is None else iterations)
    assert pr.view(np.int64).tolist() == z["rank_bits"]
    cid, csz = og.components(len(ids), s, d)
    assert cid.tolist() == z["comp_id"] and csz.tolist() == z["comp_size"]


def test_oracle_vs_live_reference_random():
    """Beyond the committed cases, where the compiled reference is available (build container)."""
    if not og.have_ref_graph():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(5)
    for trial in range(4):
        n = int(rng.integers(5, 300))
        rows = [(f"k{a}", f"k{b}") for a, b in rng.integers(0, n, (int(rng.integers(1, 4 * n)), 2))]
        damping, iters = float(rng.choice([0.5, 0.85, 0.99])), int(rng.integers(0, 25))
        ref = og.ref_graph_tvf(rows, damping, iters)
        ids, s, d = og.first_seen_edges(rows)
        pr = og.pagerank(len(ids), s, d, damping, iters)
        assert [r[0] for r in ref["pagerank"]] == ids
        assert np.array([r[1] for r in ref["pagerank"]], np.float64).view(np.int64).tolist() == pr.view(np.int64).tolist()
        cid, csz = og.components(len(ids), s, d)
        assert [r[1] for r in ref["components"]] == cid.tolist() and [r[2] for r in ref["components"]] == csz.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_device_matches_reference_golden(gpu, name):
    rows, damping, iterations = CASES[name]
    z = golden()[name]
    ids, s, d = og.first_seen_edges(rows)
    pr, st = gpu.graph.pagerank(len(ids), s, d, 0.85 if damping is None else damping, 20 if iterations is None else iterations)
    assert pr.view(np.int64).tolist() == z["rank_bits"], name  # the same f64 additions in the same order
    cid, csz, _ = gpu.graph.components(len(ids), s, d, gpu.graph.COMPONENTS_EXACT)
    assert cid.tolist() == z["comp_id"] and csz.tolist() == z["comp_size"]  # the reference's union-find roots
    fid, fsz, _ = gpu.graph.components(len(ids), s, d, gpu.graph.COMPONENTS_FAST)
    assert fsz.tolist() == z["comp_size"]
    # same partition; the parallel mode names a component by its smallest node index
    groups = {}
    for i, c in enumerate(z["comp_id"]):
        groups.setdefault(c, []).append(i)
    for members in groups.values():
        assert {int(fid[i]) for i in members} == {min(members)}


@pytest.mark.gpu
def test_device_vs_oracle_larger(gpu):
    """50k nodes / 400k edge rows with ~10 % dangling nodes, multi-edges and self loops: rank bits, roots and sizes."""
    rng = np.random.default_rng(9)
    n, m = 50_000, 400_000
    s = rng.integers(0, int(n * 0.9), m).astype(np.int32)  # the last 10 % of the ids never appear as a source
    d = rng.integers(0, n, m).astype(np.int32)
    remap = -np.ones(n, np.int64)  # first-seen renumbering, as the SQL ingest does
    order = np.stack([s, d], 1).reshape(-1)
    _, first = np.unique(order, return_index=True)
    seen = order[np.sort(first)]
    remap[seen] = np.arange(len(seen))
    s2, d2 = remap[s].astype(np.int32), remap[d].astype(np.int32)
    nn = len(seen)
    want = og.pagerank(nn, s2, d2, 0.85, 10)
    got, st = gpu.graph.pagerank(nn, s2, d2, 0.85, 10)
    assert st["dangling"] > 1000
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
    assert abs(float(got.sum()) - 1.0) < 1e-9  # rank is conserved (dangling mass is redistributed)
    wc, ws = og.components(nn, s2[:60_000], d2[:60_000])
    gc, gs, _ = gpu.graph.components(nn, s2[:60_000], d2[:60_000], gpu.graph.COMPONENTS_EXACT)
    assert np.array_equal(gc, wc) and np.array_equal(gs, ws)
    fc, fs, _ = gpu.graph.components(nn, s2, d2, gpu.graph.COMPONENTS_FAST)
    wc2, ws2 = og.components(nn, s2, d2)
    assert np.array_equal(fs, ws2)
    assert len(set(zip(fc.tolist(), wc2.tolist()))) == len(set(fc.tolist())) == len(set(wc2.tolist()))  # a bijection of ids


@pytest.mark.gpu
def test_sql_table_valued_functions_equal_the_reference(gpu, ext_conn):
    c = ext_conn
    z = golden()
    for name, (rows, damping, iterations) in sorted(CASES.items()):
        c.execute("DROP TABLE IF EXISTS e")
        c.execute("CREATE TABLE e(s TEXT, d TEXT)")
        c.executemany("INSERT INTO e VALUES (?, ?)", rows)
        extra, args = "", []
        if damping is not None:
            extra += " AND damping = ?"
            args.append(damping)
        if iterations is not None:
            extra += " AND iterations = ?"
            args.append(iterations)
        pr = c.execute("SELECT node, rank FROM graph_pagerank WHERE edge_table='e' AND src_col='s' AND dst_col='d'" + extra, args).fetchall()
        assert [r[0] for r in pr] == z[name]["nodes"]
        assert np.array([r[1] for r in pr], np.float64).view(np.int64).tolist() == z[name]["rank_bits"], name
        cc = c.execute("SELECT node, component_id, component_size FROM graph_components WHERE edge_table='e' AND src_col='s' AND dst_col='d'").fetchall()
        assert [r[0] for r in cc] == z[name]["comp_nodes"] and [r[1] for r in cc] == z[name]["comp_id"]
        assert [r[2] for r in cc] == z[name]["comp_size"]
    # the reference's error behaviour (src/graph_tvf.c:1453-1456, :1814-1817 and its un-compacted argvIndex)
    import sqlite3

    with pytest.raises(sqlite3.OperationalError, match="graph_pagerank: invalid table/column identifier"):
        c.execute("SELECT * FROM graph_pagerank WHERE edge_table='e;x' AND src_col='s' AND dst_col='d'").fetchall()
    with pytest.raises(sqlite3.OperationalError, match="graph_components: invalid table/column identifier"):
        c.execute("SELECT * FROM graph_components WHERE edge_table='e' AND src_col='s s' AND dst_col='d'").fetchall()
    with pytest.raises(sqlite3.OperationalError, match="xBestIndex malfunction"):
        c.execute("SELECT * FROM graph_pagerank WHERE edge_table='e' AND src_col='s' AND dst_col='d' AND iterations = 3").fetchall()
    c.execute("CREATE TABLE empty_e(s TEXT, d TEXT)")
    assert c.execute("SELECT * FROM graph_pagerank WHERE edge_table='empty_e' AND src_col='s' AND dst_col='d'").fetchall() == []


@pytest.mark.gpu
def test_pagerank_by_source_ranges_equals_the_oracle_and_the_one_launch_pull(gpu, monkeypatch):
    """Graphs without dangling nodes and more than 2^18 nodes are pulled one source range per launch (k_pr_pull_tile: share[]
    stays in the L2): 700k nodes (three ranges, the last one short), 4.2M rows with multi-edges, self loops, nodes with an empty
    in-list and lists that skip a whole range — rank bits against the oracle and against the one-launch kernel."""
    rng = np.random.default_rng(21)
    n, m = 700_000, 3_500_000
    s = np.concatenate([np.arange(n, dtype=np.int64), rng.integers(0, n, m)]).astype(np.int32)  # every node has an out-edge
    d = rng.integers(0, n - 5000, n + m).astype(np.int32)  # the last 5 000 nodes have no in-edge
    d[rng.integers(0, n + m, 50_000)] = 7  # one long in-list crossing every range
    lowhalf = rng.integers(0, n + m, 200_000)
    d[lowhalf] = 11  # a second hub whose sources are then cut to the first range only
    s[lowhalf] = rng.integers(0, 1 << 18, len(lowhalf)).astype(np.int32)
    s[:n] = np.arange(n, dtype=np.int32)  # (keep the out-edge of every node)
    d[:n][d[:n] == 11] = 12
    want = og.pagerank(n, s, d, 0.85, 6)
    got, st = gpu.graph.pagerank(n, s, d, 0.85, 6)
    assert st["dangling"] == 0
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
    monkeypatch.setenv("MN_PR_TILES", "0")
    one, _ = gpu.graph.pagerank(n, s, d, 0.85, 6)
    assert np.array_equal(one.view(np.int64), want.view(np.int64))


# ───────────── f-2: csr_apply_delta (the merge step of graph_adjacency's incremental rebuild) ─────────────

def _same_csr(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and ((a[2] is None and b[2] is None) or
                                                                          np.array_equal(a[2].view(np.int64), b[2].view(np.int64)))


@pytest.mark.parametrize("seed,weighted", [(1, False), (2, True), (3, False)])
def test_csr_delta_oracle_vs_live_reference(seed, weighted):
    if not og.have_ref_graph():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    case = og.delta_case(seed, weighted=weighted)
    assert _same_csr(og.csr_apply_delta(*case), og.ref_csr_apply_delta(*case))


def test_csr_delta_reference_unit_cases_on_the_oracle():
    """test/test_graph_csr.c:140-233 restated: insert A->C, delete A->B (swap with last), insert to a new node D."""
    one = lambda *v: np.array(v, np.int32)  # noqa: E731
    off, tgt, _ = og.csr_apply_delta(one(0, 1, 1, 1), one(1), None, one(0), one(2), np.ones(1), one(1), 3)
    assert off.tolist() == [0, 2, 2, 2] and tgt.tolist() == [1, 2]
    off, tgt, _ = og.csr_apply_delta(one(0, 2, 2, 2), one(1, 2), None, one(0), one(1), np.ones(1), one(2), 3)
    assert off.tolist() == [0, 1, 1, 1] and tgt.tolist() == [2]
    off, tgt, _ = og.csr_apply_delta(one(0, 1, 1), one(1), None, one(0), one(2), np.ones(1), one(1), 3)
    assert off.tolist() == [0, 2, 2, 2] and tgt.tolist() == [1, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,weighted", [(1, False), (2, True), (3, False), (4, True)])
def test_csr_delta_device_equals_oracle(gpu, seed, weighted):
    case = og.delta_case(seed, n=3000, e=40_000, nd=9000, weighted=weighted)
    assert _same_csr(gpu.graph.csr_apply_delta(*case), og.csr_apply_delta(*case))
    # empty log, empty graph, everything deleted
    off, tgt, w, dsrc, ddst, dw, dop, new_n = case
    none = np.zeros(0, np.int32)
    assert _same_csr(gpu.graph.csr_apply_delta(off, tgt, w, none, none, np.zeros(0), none, len(off) - 1),
                     og.csr_apply_delta(off, tgt, w, none, none, np.zeros(0), none, len(off) - 1))
    z = np.zeros(1, np.int32)
    got = gpu.graph.csr_apply_delta(z, none, None, np.array([0, 1], np.int32), np.array([1, 0], np.int32), np.ones(2),
                                    np.array([1, 1], np.int32), 2)
    assert got[0].tolist() == [0, 1, 2] and got[1].tolist() == [1, 0]


# ───────────── f-4: Brandes betweenness (src/graph_centrality.c:393-505) ─────────────

from oracle.graph_cases import betweenness_cases  # noqa: E402

BCASES = betweenness_cases()


def _bgolden():
    with gzip.open(os.path.join(G, "betweenness.json.gz"), "rt") as f:
        return json.load(f)


def _bcase_csr(name):
    rows, weighted, direction, normalized, approx = BCASES[name]
    ids = {}
    s, d, w = [], [], []
    for r in rows:
        for x in r[:2]:
            ids.setdefault(x, len(ids))
        s.append(ids[r[0]])
        d.append(ids[r[1]])
        w.append(r[2] if len(r) > 2 else 1.0)
    csr = og.Csr(np.array(s), np.array(d), np.array(w) if weighted else None, direction or "forward", n_nodes=len(ids),
                 first_seen=False)
    return csr, list(ids), normalized or 0, approx


def _edge_rows(csr, names, eb):
    rows, bits = [], []
    for i in range(csr.n):  # GraphData.out in list order, positive values only (src/graph_centrality.c:1172-1182)
        for e in range(csr.off_out[i], csr.off_out[i + 1]):
            t = csr.tgt_out[e]
            if eb[i, t] > 0:
                rows.append([names[i], names[t]])
                bits.append(int(eb[i, t].view(np.int64)))
    return rows, bits


@pytest.mark.parametrize("name", sorted(BCASES))
def test_betweenness_oracle_matches_reference_golden(name):
    z = _bgolden()[name]
    csr, names, normalized, approx = _bcase_csr(name)
    cb, _ = og.betweenness(csr, 50000 if approx is None else approx, normalized)  # node TVF default threshold (:838)
    assert names == z["nodes"] and cb.view(np.int64).tolist() == z["cb_bits"]
    _, eb = og.betweenness(csr, 0 if approx is None else approx, normalized, edges=True)  # edge TVF default (:1082)
    rows, bits = _edge_rows(csr, names, eb)
    assert rows == z["edges"] and bits == z["eb_bits"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(BCASES))
def test_betweenness_device_matches_reference_golden(gpu, name):
    z = _bgolden()[name]
    csr, names, normalized, approx = _bcase_csr(name)
    g = gpu.Graph(csr.n, csr.off_out, csr.tgt_out, csr.w_out if csr.weighted else None, csr.off_in, csr.tgt_in,
                  csr.w_in if csr.weighted else None)
    cb, _, _ = g.betweenness(csr.direction, 50000 if approx is None else approx, normalized)
    assert cb.view(np.int64).tolist() == z["cb_bits"], name
    _, eb, _ = g.betweenness(csr.direction, 0 if approx is None else approx, normalized, edges=True)
    rows, bits = _edge_rows(csr, names, eb)
    assert rows == z["edges"] and bits == z["eb_bits"], name
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [False, True])
def test_betweenness_device_vs_oracle_larger(gpu, weighted):
    """3 000 nodes / 15 000 edges, all sources, in several chunks of the scratch budget: node values bit-identical."""
    rng = np.random.default_rng(12)
    n, m = 3000, 15000
    s, d = rng.integers(0, n, m), rng.integers(0, n, m)
    w = rng.integers(1, 5, m).astype(np.float64) if weighted else None
    csr = og.Csr(s, d, w, "both", n_nodes=n, first_seen=False)
    want, _ = og.betweenness(csr, 0, 1)
    g = gpu.Graph(csr.n, csr.off_out, csr.tgt_out, csr.w_out if weighted else None, csr.off_in, csr.tgt_in, csr.w_in if weighted else None)
    os.environ["MN_BRANDES_SCRATCH_MB"] = "256"  # several chunks of sources: the source-order accumulation crosses chunks
    try:
        got, _, ms = g.betweenness("both", 0, 1)
        os.environ["MN_BRANDES_LANES"] = "64"  # full wavefronts (the default narrows them until a launch has thousands)
        wide, _, _ = g.betweenness("both", 0, 1)
    finally:
        os.environ.pop("MN_BRANDES_SCRATCH_MB")
        os.environ.pop("MN_BRANDES_LANES", None)
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
    assert np.array_equal(wide.view(np.int64), want.view(np.int64))
    one, _, _ = g.betweenness("both", 0, 1)  # the default budget: every source in one launch
    assert np.array_equal(one.view(np.int64), want.view(np.int64))
    g.close()


@pytest.mark.gpu
def test_betweenness_sql_equals_the_reference(gpu, ext_conn):
    c = ext_conn
    z = _bgolden()
    for name, (rows, weighted, direction, normalized, approx) in sorted(BCASES.items()):
        c.execute("DROP TABLE IF EXISTS e")
        c.execute("CREATE TABLE e(s TEXT, d TEXT, w REAL)")
        c.executemany("INSERT INTO e VALUES (?, ?, ?)", [(r[0], r[1], r[2] if len(r) > 2 else None) for r in rows])
        extra, args = "", []
        for col, val in (("weight_col", "w" if weighted else None), ("direction", direction), ("normalized", normalized),
                         ("auto_approx_threshold", approx)):
            if val is not None:
                extra += f" AND {col} = ?"
                args.append(val)
        where = "edge_table='e' AND src_col='s' AND dst_col='d'" + extra
        nodes = c.execute("SELECT node, centrality FROM graph_node_betweenness WHERE " + where, args).fetchall()
        assert [r[0] for r in nodes] == z[name]["nodes"], name
        assert np.array([r[1] for r in nodes], np.float64).view(np.int64).tolist() == z[name]["cb_bits"], name
        edges = c.execute("SELECT src, dst, centrality FROM graph_edge_betweenness WHERE " + where, args).fetchall()
        assert [[r[0], r[1]] for r in edges] == z[name]["edges"], name
        assert np.array([r[2] for r in edges], np.float64).view(np.int64).tolist() == z[name]["eb_bits"], name


@pytest.mark.gpu
@pytest.mark.parametrize("n,deg", [(10_000, 20), (30_000, 3), (90_000, 4), (500, 1)])
def test_components_exact_root_ids_by_blocks_equal_the_reference_order(gpu, n, deg):
    """MN_COMPONENTS_EXACT (round 4): 1 024 rows are tested at once for "already in one tree" and only the others are replayed in
    row order — the union-find ROOT ids must still be the ones the reference's row-by-row loop ends with (src/graph_tvf.c:
    1249-1273, 1314-1360).  10 000 / 30 000 nodes run with the table in LDS, 90 000 in global memory; sparse graphs keep many
    components and many effective unions, duplicates and self loops included."""
    import bench_graph as bg

    s, d = bg.er_rows(n, deg, seed=n)
    extra = np.random.default_rng(n).integers(0, n, 200).astype(np.int32)
    s = np.concatenate([s, extra, extra[:50]]).astype(np.int32)  # repeated rows and self loops
    d = np.concatenate([d, extra[::-1], extra[:50]]).astype(np.int32)
    want_id, want_sz = og.components(n, s, d)
    got_id, got_sz, st = gpu.graph.components(n, s, d, gpu.graph.COMPONENTS_EXACT)
    assert np.array_equal(got_id, want_id) and np.array_equal(got_sz, want_sz)
