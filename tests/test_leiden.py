"""Leiden (src/graph_community.c): oracle vs the reference's golden communities/Q (CPU), HIP sequential
mode vs the same golden (GPU, bit-exact), HIP batched mode vs the oracle's restatement of the same
schedule (GPU, bit-exact) and within tolerance of the sequential modularity."""
import os

import numpy as np
import pytest

from oracle import orc_graph as og
from oracle.graph_cases import er, leiden_cases, planted

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = leiden_cases()


def qbits(q):
    return np.array([q], np.float64).view(np.int64)[0]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_golden(name):
    z = np.load(os.path.join(G, "leiden.npz"))
    s, d, w, res = CASES[name]
    comm, q, st = og.leiden(og.Csr(s, d, w, "both"), res, 1)
    assert np.array_equal(comm, z[f"{name}_community"])
    assert qbits(q) == z[f"{name}_q"][0]


@pytest.mark.parametrize("name", ["karate", "er200", "er2000", "er2000w", "planted600"])
@pytest.mark.parametrize("batch", [16, 1000, -3])
def test_batched_schedule_pin(name, batch):
    """The batch-synchronous schedule is this repository's own (the reference has no such mode), so nothing in the
    reference pins it: this file does (oracle/gen_golden.py leiden_schedule_pins).  A change of round sizes, commit rule
    or tail rule shows up here, on CPU, and has to be made on purpose on both sides (device vs live oracle: -m gpu)."""
    z = np.load(os.path.join(G, "leiden_schedule_pins.npz"))
    s, d, w, res = CASES[name]
    comm, q, st = og.leiden(og.Csr(s, d, w, "both"), res, batch)
    key = f"{name}_b{batch}"
    assert np.array_equal(comm, z[key + "_community"])
    assert qbits(q) == z[key + "_q"][0]
    assert [st["move_sweeps"], st["refine_sweeps"], st["moves"]] == z[key + "_sweeps"].tolist()


def test_reference_structural_contracts():
    """pytests/test_graph_community.py:129-287 restated on the oracle: barbell → {0,1,2},{3,4,5}; triangle → 1;
    disconnected → 2; Q > 0; ids contiguous from 0; resolution monotone."""
    comm, q, _ = og.leiden(og.Csr(*CASES["barbell"][:3], "both"), 1.0, 1)
    assert len(set(comm[:3])) == 1 and len(set(comm[3:])) == 1 and comm[0] != comm[3] and q > 0
    comm, q, _ = og.leiden(og.Csr(*CASES["triangle"][:3], "both"), 1.0, 1)
    assert len(set(comm)) == 1
    comm, q, _ = og.leiden(og.Csr(*CASES["disconnected"][:3], "both"), 1.0, 1)
    assert len(set(comm)) == 2
    s, d, w, _ = CASES["karate"]
    k_lo = len(set(og.leiden(og.Csr(s, d, w, "both"), 0.5, 1)[0]))
    k_hi = len(set(og.leiden(og.Csr(s, d, w, "both"), 2.0, 1)[0]))
    assert k_lo <= k_hi
    comm = og.leiden(og.Csr(s, d, w, "both"), 1.0, 1)[0]
    assert sorted(set(comm)) == list(range(comm.max() + 1))


def test_live_reference_random_graphs():
    if not og.have_ref_graph():
        pytest.skip("compiled reference not present")
    for seed in range(5):
        s, d, w = er(300 + 50 * seed, 1500, 100 + seed, weighted=seed % 2 == 1)
        rc, rq, _ = og.ref_leiden(s, d, w, "both", 1.0)
        oc, oq, _ = og.leiden(og.Csr(s, d, w, "both"), 1.0, 1)
        assert np.array_equal(rc, oc) and qbits(rq) == qbits(oq)


def test_batched_schedule_quality_on_cpu():
    s, d, w = planted(600, 6, 0.15, 0.005, 7)
    csr = og.Csr(s, d, w, "both")
    _, q_seq, _ = og.leiden(csr, 1.0, 1)
    for batch in (16, 256, 100000, -2, -3, -4):  # rounds of `batch` nodes; < 0: whole-graph synchronous sweeps (pick-less period)
        comm, q, st = og.leiden(csr, 1.0, batch)
        assert q > 0 and q >= 0.85 * q_seq, (batch, q, q_seq)
        assert sorted(set(comm)) == list(range(comm.max() + 1))


# ───────────────────────── GPU ─────────────────────────

def _dev_graph(gpu, csr):
    return gpu.Graph(csr.n, csr.off_out, csr.tgt_out, csr.w_out if csr.weighted else None, csr.off_in, csr.tgt_in,
                     csr.w_in if csr.weighted else None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_sequential_matches_reference_golden(gpu, name):
    z = np.load(os.path.join(G, "leiden.npz"))
    s, d, w, res = CASES[name]
    g = _dev_graph(gpu, og.Csr(s, d, w, "both"))
    comm, q, st = g.leiden(res, "both", gpu.LEIDEN_SEQUENTIAL)
    assert np.array_equal(comm, z[f"{name}_community"]), name
    assert qbits(q) == z[f"{name}_q"][0]
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["karate", "er200", "er2000", "er2000w", "planted600"])
@pytest.mark.parametrize("batch", [16, 1000, 65536])
def test_gpu_batched_matches_oracle_schedule(gpu, name, batch):
    s, d, w, res = CASES[name]
    csr = og.Csr(s, d, w, "both")
    oc, oq, ost = og.leiden(csr, res, batch)
    g = _dev_graph(gpu, csr)
    comm, q, st = g.leiden(res, "both", gpu.LEIDEN_BATCHED, batch)
    assert np.array_equal(comm, oc), (name, batch)
    assert qbits(q) == qbits(oq)
    assert st["moves"] == ost["moves"]
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("batch", [0, -2, -4])
def test_gpu_synchronous_sweeps_match_oracle_schedule(gpu, name, batch):
    """The default schedule (batch 0 = whole-graph synchronous sweeps, pick-less every 3rd) and two other periods against the
    oracle's restatement (sync_phase): communities, Q bits, moves and sweep counts.  The Erdos-Renyi cases do not settle
    within the sweep cap and are finished by the round schedule — that hand-over is part of what is compared."""
    s, d, w, res = CASES[name]
    csr = og.Csr(s, d, w, "both")
    oc, oq, ost = og.leiden(csr, res, batch if batch else -3)
    g = _dev_graph(gpu, csr)
    comm, q, st = g.leiden(res, "both", gpu.LEIDEN_BATCHED, batch)
    assert np.array_equal(comm, oc), (name, batch)
    assert qbits(q) == qbits(oq)
    assert (st["moves"], st["move_sweeps"], st["refine_sweeps"]) == (ost["moves"], ost["move_sweeps"], ost["refine_sweeps"])
    g.close()


@pytest.mark.gpu
def test_gpu_high_degree_nodes_use_global_scratch(gpu):
    """a hub with > 1024 edges exercises the global-scratch path of best_move"""
    n = 3000
    hub_s = np.zeros(n - 1, np.int32)
    hub_d = np.arange(1, n, dtype=np.int32)
    s2, d2, _ = er(n, 6000, 9)
    # two more hubs, one in the first hub's round and one in a later round: each has its own scratch region
    h1_d = np.arange(2, 1502, dtype=np.int32)
    h2_d = np.arange(100, 1300, dtype=np.int32)
    s = np.concatenate([hub_s, s2, np.full(len(h1_d), 1, np.int32), np.full(len(h2_d), 2500, np.int32)])
    d = np.concatenate([hub_d, d2, h1_d, h2_d])
    csr = og.Csr(s, d, None, "both")
    oc, oq, _ = og.leiden(csr, 1.0, 1)
    g = _dev_graph(gpu, csr)
    comm, q, _ = g.leiden(1.0, "both", gpu.LEIDEN_SEQUENTIAL)
    assert np.array_equal(comm, oc) and qbits(q) == qbits(oq)
    bc, bq, _ = og.leiden(csr, 1.0, 512)
    comm, q, _ = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED, 512)
    assert np.array_equal(comm, bc) and qbits(q) == qbits(bq)
    sc, sq, _ = og.leiden(csr, 1.0, -3)
    comm, q, _ = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED)  # default: synchronous sweeps (one launch over all nodes, hubs included)
    assert np.array_equal(comm, sc) and qbits(q) == qbits(sq)
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [False, True])
def test_gpu_batched_matches_oracle_schedule_at_100k_nodes(gpu, weighted):
    """config 5's kernels at a size between the unit graphs and the bench: LFR-like, 100k nodes / ~1M edges, rounds of
    4096 nodes; unweighted = the O(degree) hashed evaluation + device-side bookkeeping, weighted = list-order f64 sums +
    host bookkeeping.  Communities, Q bits and the move count equal the CPU restatement of the same schedule."""
    s, d, _ = gpu.lfr.lfr_like(100_000, 20, 100, 0.3, seed=5)
    w = (np.random.default_rng(8).random(len(s)) * 2 + 0.5) if weighted else None
    csr = og.Csr(s, d, w, "both", n_nodes=100_000, first_seen=False)
    oc, oq, ost = og.leiden(csr, 1.0, 4096)
    g = _dev_graph(gpu, csr)
    comm, q, st = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED, 4096)
    assert np.array_equal(comm, oc) and qbits(q) == qbits(oq) and st["moves"] == ost["moves"]
    comm2, q2, _ = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED, 4096)  # the per-graph workspace is reused: same answer again
    assert np.array_equal(comm2, oc) and qbits(q2) == qbits(oq)
    # the default schedule (whole-graph synchronous sweeps) on the same graph, same workspace
    sc, sq, sst = og.leiden(csr, 1.0, -3)
    comm3, q3, st3 = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED)
    assert np.array_equal(comm3, sc) and qbits(q3) == qbits(sq) and st3["moves"] == sst["moves"]
    assert (st3["move_sweeps"], st3["refine_sweeps"]) == (sst["move_sweeps"], sst["refine_sweeps"])
    g.close()


@pytest.mark.gpu
def test_config5_full_size_properties(gpu):
    """BASELINE config 5 at full size (LFR-like, 500k nodes / ~9.3M edges, 1 GPU): size-independent properties of run_leiden's
    output — contiguous ids in first-seen order, Q = compute_modularity of the returned partition recomputed on the host in
    f64 (1e-9), the planted communities recovered (NMI > 0.99), and the same bits when run again on the reused workspace."""
    n = 500_000
    s, d, truth = gpu.lfr.lfr_like(n, 40, 200, 0.3)
    g = gpu.graph.graph_from_edges(n, s, d)
    comm, q, st = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED)
    comm2, q2, _ = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED)
    g.close()
    assert np.array_equal(comm, comm2) and qbits(q) == qbits(q2)
    K = int(comm.max()) + 1
    first = np.full(K, n, np.int64)
    np.minimum.at(first, comm, np.arange(n))
    assert (np.diff(first) > 0).all()  # renumbered in first-seen order (src/graph_community.c:317-331)
    # modularity of the partition, both directions (every undirected edge appears in out[] of one end and in[] of the other)
    m = float(len(s))
    deg = np.bincount(s, minlength=n) + np.bincount(d, minlength=n)
    tot = np.bincount(comm, weights=deg, minlength=K)
    inside = np.bincount(comm[s][comm[s] == comm[d]], minlength=K) * 2.0
    q_host = float(np.sum(inside / (2 * m) - (tot / (2 * m)) ** 2))
    assert abs(q - q_host) < 1e-9 and q > 0.6
    a = np.unique(truth, return_inverse=True)[1]
    cont = np.zeros((a.max() + 1, K))
    np.add.at(cont, (a, comm), 1)
    pa, pb, pab = cont.sum(1) / n, cont.sum(0) / n, cont / n
    nz = pab > 0
    mi = (pab[nz] * np.log(pab[nz] / (pa[:, None] * pb[None, :])[nz])).sum()
    nmi = 2 * mi / (-(pa[pa > 0] * np.log(pa[pa > 0])).sum() - (pb[pb > 0] * np.log(pb[pb > 0])).sum())
    assert nmi > 0.99 and st["move_sweeps"] > 0
