"""The C-ABI's exception barrier (csrc/mn_guard.hpp; the reference's convention under memory pressure, src/hnsw_algo.h:55-79):
mn_debug_fault_alloc(n) makes the n-th host allocation of the library throw std::bad_alloc.  For every n a call makes, the call
must come back with its error value and a message — never a C++ exception through the extern "C" frame (which would abort
the process: the test run itself is the witness) — and leave its handle either usable with the right contents or cleanly
unusable ("index unusable"), and a statement through the SQLite extension must fail with an ordinary SQLite error."""
import sqlite3

import numpy as np
import pytest

from oracle import orc_graph as og
from oracle.graph_cases import leiden_cases
from util import gauss

pytestmark = pytest.mark.gpu


def _walk(L, op, max_points=24):
    """allocation counts at which to fail: every one when the call makes few, an even spread otherwise"""
    L.mn_debug_fault_alloc(0)
    op()
    total = L.mn_debug_fault_alloc(0)
    if total <= max_points:
        return list(range(1, total + 1)), total
    return sorted(set(int(x) for x in np.linspace(1, total, max_points))), total


def _fresh(gpu, n=400, dim=16, M=8, efc=40):
    X = gauss(n, dim, 21)
    ids = np.arange(1, n + 1, dtype=np.int64)
    g = gpu.HnswIndex(dim, "l2", M, efc)
    assert g.insert_batch(ids, X, gpu.BUILD_BATCHED) == 0
    return g, X, ids


@pytest.mark.parametrize("what", ["insert_exact", "insert_batch", "insert_many_exact", "search_batch", "search_one", "delete", "build"])
def test_hnsw_calls_fail_cleanly_at_every_host_allocation(gpu, what):
    L = gpu.lib()
    dim = 16
    Xn = gauss(64, dim, 22)
    idn = np.arange(10_001, 10_065, dtype=np.int64)
    Q = gauss(8, dim, 23)

    def call(g):
        if what == "insert_exact":
            return g.insert_batch(idn[:1], Xn[:1], gpu.BUILD_SEQUENTIAL)
        if what == "insert_many_exact":  # speculative windows
            return g.insert_batch(idn[:24], Xn[:24], gpu.BUILD_SEQUENTIAL)
        if what == "insert_batch":
            return g.insert_batch(idn, Xn, gpu.BUILD_BATCHED)
        if what == "build":
            return g.build(idn, Xn, 16, 8192)
        if what == "delete":
            return g.delete(7)
        if what == "search_one":
            return 0 if len(g.search(Q[0], 5, 32)[0]) == 5 else -1
        try:
            g.search_batch(Q, 5, 32)
            return 0
        except gpu.hnsw.MuninnHipError:
            return -1

    g, X, ids = _fresh(gpu)
    points, total = _walk(L, lambda: call(g))
    g.close()
    if total == 0:  # e.g. a small batched search: pinned staging block, no host allocation at all — nothing can throw
        return
    failed = usable_after = broken_after = 0
    for nth in points:
        g, X, ids = _fresh(gpu)
        want_i, want_d, _ = g.search_batch(Q, 5, 32)
        L.mn_debug_fault_alloc(0)
        L.mn_debug_fault_alloc(nth)
        rc = call(g)
        L.mn_debug_fault_alloc(0)
        if rc == 0:  # (counts vary a little between runs: the nth allocation did not happen this time)
            g.close()
            continue
        failed += 1
        msg = gpu.hnsw._err()
        assert msg, (what, nth)
        # the handle: either it still answers exactly as before the failed call (mutations were taken back) ...
        try:
            got_i, got_d, _ = g.search_batch(Q, 5, 32)
            if what in ("search_batch", "search_one"):
                assert np.array_equal(got_i, want_i) and np.array_equal(got_d.view(np.int32), want_d.view(np.int32)), (what, nth)
            if what.startswith("insert") or what == "build":
                # ... and takes the same rows now (nothing of the failed call stayed behind), or it says it is unusable
                rc2 = call(g)
                assert rc2 == 0 or "unusable" in gpu.hnsw._err(), (what, nth, gpu.hnsw._err())
            usable_after += 1
        except gpu.hnsw.MuninnHipError as e:
            assert "unusable" in str(e), (what, nth, str(e))
            broken_after += 1
        g.close()
    assert failed >= max(1, len(points) // 2), (what, failed, points)
    print(f"{what}: {total} host allocations, {len(points)} injection points, {failed} failed calls, "
          f"{usable_after} handles usable / {broken_after} cleanly unusable afterwards")


def test_exact_insert_that_fails_on_memory_leaves_the_graph_the_reference_would_have(gpu, orc):
    """an exact insert that fails before any link row is rewritten is taken back: the next insert of the same row gives the
    oracle's (= the reference's) graph, level stream included"""
    L = gpu.lib()
    n, dim = 300, 12
    X = gauss(n + 1, dim, 31)
    ids = np.arange(1, n + 2, dtype=np.int64)
    o = orc.Oracle(dim, "l2", 8, 40)
    for i in range(n + 1):
        assert o.insert(int(ids[i]), X[i]) == 0
    g = gpu.HnswIndex(dim, "l2", 8, 40)
    assert g.insert_batch(ids[:n], X[:n], gpu.BUILD_SEQUENTIAL) == 0
    L.mn_debug_fault_alloc(0)
    L.mn_debug_fault_alloc(1)  # the very first allocation of the call: nothing has changed yet
    assert g.insert_batch(ids[n:], X[n:], gpu.BUILD_SEQUENTIAL) == -1
    L.mn_debug_fault_alloc(0)
    assert "memory" in gpu.hnsw._err()
    assert g.insert_batch(ids[n:], X[n:], gpu.BUILD_SEQUENTIAL) == 0
    assert g.graph(ids) == o.graph(ids)
    g.close()


def _dev_graph(gpu, csr):
    return gpu.Graph(csr.n, csr.off_out, csr.tgt_out, csr.w_out if csr.weighted else None, csr.off_in, csr.tgt_in,
                     csr.w_in if csr.weighted else None)


@pytest.mark.parametrize("name", ["er2000", "er2000w"])
@pytest.mark.parametrize("mode", ["sequential", "batched"])
def test_leiden_fails_cleanly_and_the_graph_stays_usable(gpu, name, mode):
    L = gpu.lib()
    s, d, w, res = leiden_cases()[name]
    csr = og.Csr(s, d, w, "both")
    g = _dev_graph(gpu, csr)
    m = gpu.LEIDEN_SEQUENTIAL if mode == "sequential" else gpu.LEIDEN_BATCHED
    want = g.leiden(res, "both", m)
    points, total = _walk(L, lambda: g.leiden(res, "both", m))
    failed = 0
    for nth in points:
        L.mn_debug_fault_alloc(0)
        L.mn_debug_fault_alloc(nth)
        try:
            g.leiden(res, "both", m)
        except gpu.graph.MuninnHipError as e:
            failed += 1
            assert "memory" in str(e) or "exception" in str(e), str(e)
        finally:
            L.mn_debug_fault_alloc(0)
        got = g.leiden(res, "both", m)  # the same handle, the same answer
        assert np.array_equal(got[0], want[0]) and got[1] == want[1], (name, mode, nth)
    g.close()
    assert total == 0 or failed >= 1


def test_node2vec_train_fails_cleanly(gpu):
    L = gpu.lib()
    from oracle.graph_cases import er

    s, d, _ = er(400, 2000, 5)
    off, adj = gpu.graph.n2v_csr_from_edges(400, s, d)
    prm = dict(p=1.0, q=1.0, num_walks=2, walk_length=10, window=3, neg_samples=2, learning_rate=0.025, epochs=1)
    for mode in (gpu.N2V_SEQUENTIAL, gpu.N2V_BATCHED):
        want, _ = gpu.node2vec_train(off, adj, 16, mode=mode, **prm)
        points, total = _walk(L, lambda: gpu.node2vec_train(off, adj, 16, mode=mode, **prm), max_points=16)
        failed = 0
        for nth in points:
            L.mn_debug_fault_alloc(0)
            L.mn_debug_fault_alloc(nth)
            try:
                gpu.node2vec_train(off, adj, 16, mode=mode, **prm)
            except gpu.graph.MuninnHipError as e:
                failed += 1
                assert str(e), "a failed call names its failure"
            finally:
                L.mn_debug_fault_alloc(0)
        got, _ = gpu.node2vec_train(off, adj, 16, mode=mode, **prm)
        assert np.array_equal(got.view(np.int32), want.view(np.int32))
        assert total == 0 or failed >= 1


def test_sql_statement_fails_with_a_sqlite_error_not_an_abort(ext_conn, gpu, monkeypatch):
    """src/hnsw_vtab.c:749-752: a failed hnsw_insert is "insert failed", SQLITE_ERROR — through this extension too, when the
    device shim runs out of host memory inside the INSERT or the SELECT."""
    import struct

    L = gpu.lib()
    monkeypatch.setenv("MUNINN_HNSW_MODE", "exact")
    c = ext_conn
    c.execute("CREATE VIRTUAL TABLE t USING hnsw_index(dimensions=4, metric='l2', m=4, ef_construction=16)")
    rng = np.random.default_rng(3)
    vec = lambda v: struct.pack("<4f", *v)
    for i in range(1, 60):
        c.execute("INSERT INTO t(rowid, vector) VALUES (?, ?)", (i, vec(rng.standard_normal(4))))
    c.commit()
    q = vec([0.1, 0.2, 0.3, 0.4])
    want = c.execute("SELECT rowid FROM t WHERE vector MATCH ? AND k = 5", (q,)).fetchall()
    errors = 0
    for nth in (1, 2, 3, 5, 8):
        L.mn_debug_fault_alloc(0)
        L.mn_debug_fault_alloc(nth)
        try:
            c.execute("INSERT INTO t(rowid, vector) VALUES (?, ?)", (1000 + nth, vec(rng.standard_normal(4))))
            c.commit()
        except sqlite3.Error as e:
            errors += 1  # (an ordinary SQLite error, with the shim's message where the statement has one to carry it)
            c.rollback()
        finally:
            L.mn_debug_fault_alloc(0)
    assert errors >= 1
    for nth in (1, 2, 3):
        L.mn_debug_fault_alloc(0)
        L.mn_debug_fault_alloc(nth)
        try:
            got = c.execute("SELECT rowid FROM t WHERE vector MATCH ? AND k = 5", (q,)).fetchall()
            assert got == [] or len(got) == 5  # hnsw_search's convention: 0 results on failure
        except sqlite3.Error:
            pass
        finally:
            L.mn_debug_fault_alloc(0)
    # the connection and the table are still there: either they answer, or they say why not — the process did not abort
    try:
        c.execute("SELECT rowid FROM t WHERE vector MATCH ? AND k = 5", (q,)).fetchall()
    except sqlite3.Error as e:
        assert "unusable" in str(e)
