#!/usr/bin/env python3
"""Graph half of the hot path on one MI355X: BASELINE.json configs[3] (Node2Vec) and configs[4] (Leiden).

bench.py measures the headline metric (kNN search).  This script applies the same contract to the two graph
workloads — one JSON line each, with `roofline` (algorithmic bytes per SURVEY §8(d) ÷ device time from HIP events)
and `cpu_baseline` (the CPU restatement of the reference algorithm, one thread, bounded sample) — so that rows
a14–a21 of SURVEY §8 are measured the same way as a1–a13.

  python bench_graph.py --workload node2vec     # 1M nodes / 20M-edge ER, p=q=1, dim 128, 10 walks x 80
  python bench_graph.py --workload leiden       # LFR-like, n=500k, <k>=40 (≈10M edges), mu=0.3
  python bench_graph.py                         # both

A step is one complete run (node2vec_train's compute / run_leiden) on an HBM-resident CSR; CSR upload and the
embedding download are outside the timed device interval (`device_ms`) but inside `ms_per_step`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0
from bench import host_cpu  # noqa: E402  (model name and core counts of the host the CPU baselines ran on)


def _traffic(key):
    """HBM-side bytes of one run from the committed PMC passes (profiles/traffic.json), when this workload was profiled and
    the kernel's sources are still the ones it was measured on (bench.py kernel_sources_sha)."""
    from bench import kernel_sources_sha

    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]
    except (OSError, ValueError, KeyError):
        return None
    if not ent.get("kernel_sources_sha256") or kernel_sources_sha(ent.get("kernel_sources", [])) != ent["kernel_sources_sha256"]:
        return None
    return ent["traffic_bytes"]


def er_edges(n, m, seed=42):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, n, m)
    d = rng.integers(0, n, m)
    keep = s != d
    return s[keep], d[keep]


def ba_edges(n, m, seed=42):
    """Barabasi-Albert preferential attachment as the reference harness builds it (benchmarks/harness/common.py:689-742: a
    complete graph on the first m + 1 nodes, then every new node attaches to m nodes drawn in proportion to their degree),
    vectorised: nodes arrive in blocks of 1/8 of the nodes present, a block draws its targets from the ends of the edges that
    existed when the block started (duplicate targets of one node are dropped by the adjacency build, as a set is there)."""
    rng = np.random.default_rng(seed)
    iu, ju = np.triu_indices(m + 1, 1)
    src, dst = [ju.astype(np.int64)], [iu.astype(np.int64)]
    ends = [src[0], dst[0]]
    n_ends = 2 * len(iu)
    pool = np.concatenate(ends)
    v = m + 1
    while v < n:
        b = min(max(1, v // 8), n - v)
        new = np.repeat(np.arange(v, v + b, dtype=np.int64), m)
        tg = pool[rng.integers(0, n_ends, b * m)]
        src.append(new)
        dst.append(tg)
        pool = np.concatenate([pool, new, tg])
        n_ends = len(pool)
        v += b
    return np.concatenate(src), np.concatenate(dst)


def _dist_ctx(args):
    """(rank, world, dist module or None, device ordinal) — ranks are started by torch.distributed.run (see bench.py)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, None, max(args.device, 0)
    import torch
    import torch.distributed as dist

    dev = args.device if args.device >= 0 else (local % max(1, torch.cuda.device_count()) if args.backend == "gloo" else local)
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    return rank, world, dist, dev


def _max_over_ranks(dist, args, x):
    if dist is None:
        return x
    import torch

    t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_node2vec(pkg, args):
    n, m, dim = args.n2v_nodes, args.n2v_edges, 128
    quick = getattr(args, "quick", False)  # bench.py's `graph` block: one training run, no host-side index build
    rank, world, dist, dev = args.ctx
    prm = dict(p=1.0, q=1.0, num_walks=10, walk_length=80, window=5, neg_samples=5, learning_rate=0.025, epochs=1)
    t0 = time.perf_counter()
    model = getattr(args, "n2v_model", "er")
    off, adj = pkg.graph.n2v_csr_from_edges(n, *(ba_edges(n, max(1, m // n)) if model == "ba" else er_edges(n, m)))
    gen_s = time.perf_counter() - t0
    deg = np.diff(off)
    if args.dump_csr:  # for tools/n2v_bench.cpp (the same graph under rocprofv3 --pmc, without Python in the process)
        with open(args.dump_csr, "wb") as f:
            f.write(np.int32(n).tobytes() + np.int64(len(adj)).tobytes())
            f.write(np.ascontiguousarray(off, np.int32).tobytes())
            f.write(np.ascontiguousarray(adj, np.int32).tobytes())
        if args.dump_only:
            return None

    # parity on a graph the CPU finishes in a second: batch-synchronous schedule, device vs its CPU restatement
    from oracle import orc_graph as og

    ps, pd = er_edges(3000, 20000, seed=7)
    ps = np.concatenate([np.arange(2999), ps])  # a chain first: the reference numbers nodes in first-seen order
    pd = np.concatenate([np.arange(1, 3000), pd])
    poff, padj = pkg.graph.n2v_csr_from_edges(3000, ps, pd)
    small = dict(prm, num_walks=2, walk_length=20)
    pe, pst = pkg.node2vec_train(poff, padj, 32, mode=pkg.N2V_BATCHED, batch_walks=512, **small)  # also warms the kernels
    og_g = og.N2vGraph(ps, pd)
    oe, opairs = og.node2vec_train_batched(og_g, 32, 1.0, 1.0, 2, 20, 5, 5, 0.025, 1, 512)
    parity = {"graph": "chain + ER, 3000 nodes / 23000 edges, dim 32, 2 walks x 20, batch 512",
              "embedding_bits_identical": bool(og_g.n == 3000 and np.array_equal(pe.view(np.int32), oe.view(np.int32))),
              "pairs_identical": bool(pst["pairs"] == opairs)}

    def train():
        if world > 1:  # config 4: walks of every batch split over the ranks, samples all-gathered (RCCL), replicas apply
            return pkg.parallel.node2vec_train_distributed(off, adj, dim, device=dev, **prm)
        return pkg.node2vec_train(off, adj, dim, mode=pkg.N2V_BATCHED, device=dev, **prm)

    def barrier():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        train()
    walls, devs, pairs = [], [], 0
    for _ in range(args.steps):
        barrier()
        t0 = time.perf_counter()
        emb, st = train()
        barrier()
        walls.append(_max_over_ranks(dist, args, time.perf_counter() - t0))
        devs.append(_max_over_ranks(dist, args, st["device_ms"]))
        pairs = st["pairs"]
    wall, dev_ms = float(np.mean(walls)), float(np.mean(devs))
    if rank != 0:
        return None
    # config 4's last leg, "-> hnsw0 index" (src/node2vec.c:540-583 INSERTs every embedding into an hnsw_index): the index
    # built from the embeddings (a) as the C-ABI's host path does it — embeddings down to the host, back up inside
    # mn_hnsw_build — and (b) device-resident: mn_node2vec_train_into trains and builds without the embeddings leaving HBM
    to_index = None
    if world == 1 and not args.no_index_leg:
        ids = np.arange(1, n + 1, dtype=np.int64)
        host_build_s = None
        if not quick:
            ixh = pkg.HnswIndex(dim, "cosine", 16, 200, device=dev)
            t0 = time.perf_counter()
            if ixh.build(ids, emb, 16, 8192) != 0:
                raise SystemExit("index build failed: " + pkg.hnsw._err())
            ixh.sync()
            host_build_s = time.perf_counter() - t0
            ixh.close()
        ixd = pkg.HnswIndex(dim, "cosine", 16, 200, device=dev)
        t0 = time.perf_counter()
        _, sti = pkg.graph.node2vec_train_into(off, adj, dim, ixd, 1, False, **prm)
        total_s = time.perf_counter() - t0
        bst = ixd.build_stats()
        ixd.close()
        # the index build's own roofline: its search half (k_beam<BUILD>) by SURVEY §8(d)'s bytes, counted on the device
        ab = bst["n_dist"] * dim * 4 + bst["n_expanded"] * 32 * 4 + bst["n_dist"] * 4
        to_index = {"index": f"hnsw_index {n} x {dim} cosine M=16 efC=200, batch-synchronous build",
                    "via_host_index_build_s": host_build_s, "via_host_total_s": None if host_build_s is None else wall + host_build_s,
                    "device_resident_index_build_s": sti["build_seconds"], "device_resident_total_s": total_s,
                    "index_vectors_per_s_device_resident": n / max(sti["build_seconds"], 1e-9),
                    "n_dist_per_insert": bst["n_dist"] / max(1, bst["nodes"]),
                    "build_roofline": {"bound": "hbm", "kernel": "k_beam<BUILD>", "kernel_ms_total": bst["search_ms"],
                                       "link_ms_total": bst["link_ms"], "batches": bst["batches"], "algorithmic_bytes": ab,
                                       "achieved": ab / max(bst["search_ms"], 1e-9) / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": ab / max(bst["search_ms"], 1e-9) / 1e6 / HBM_PEAK_GBS}}
    # SGNS reads+writes (1+neg) context rows and the centre row per pair: (2(1+neg)+2)·dim·4 B;
    # the walk adds deg(cur)·4 B per step
    steps_walk = n * prm["num_walks"] * (prm["walk_length"] - 1)
    alg = pairs * (2 * (1 + prm["neg_samples"]) + 2) * dim * 4 + steps_walk * (len(adj) / n) * 4
    achieved = alg / (dev_ms * 1e-3) / 1e9

    # CPU: the reference's own node2vec_train (oracle/_ref/muninn.so through SQL, as its users call it; output into a plain
    # table so that no HNSW insert is timed) and the serial restatement of its walk + SGNS loop (oracle/mn_graph_oracle.c), same
    # parameters, on an ER graph with the same mean degree but few enough nodes for ~10-20 s of CPU work (the reference looks
    # node names up linearly — O(N^2) — and cannot run config 4 itself)
    cn = args.n2v_cpu_nodes
    cs, cd = er_edges(cn, int(m * (cn / n)), seed=42)
    cg = og.N2vGraph(cs, cd)
    t0 = time.perf_counter()
    cemb, cpairs = og.node2vec_train(cg, dim, 1.0, 1.0, prm["num_walks"], prm["walk_length"], 5, 5, 0.025, 1)
    cpu_s = time.perf_counter() - t0
    cpu = {"value": cpairs / cpu_s, "unit": "pairs/s", "cores": 1, "kind": "port", "host": host_cpu(),
           "sample": f"serial walk+SGNS restatement (oracle/mn_graph_oracle.c) on an ER graph of {cn} nodes with "
                     f"the same mean degree and parameters: {cpairs} pairs in {cpu_s:.1f}s"}
    ref_so = os.path.join(ROOT, "oracle", "_ref", "muninn")
    if os.path.exists(ref_so + ".so"):
        import sqlite3

        c = sqlite3.connect(":memory:")
        c.enable_load_extension(True)
        c.load_extension(ref_so)
        c.execute("CREATE TABLE e(src TEXT, dst TEXT)")
        c.executemany("INSERT INTO e VALUES (?,?)", [(f"n{a}", f"n{b}") for a, b in zip(cs, cd)])
        c.execute("CREATE TABLE o(vector BLOB)")
        t0 = time.perf_counter()
        rows = c.execute("SELECT node2vec_train('e','src','dst','o',?,1.0,1.0,?,?,5,5,0.025,1)",
                         (dim, prm["num_walks"], prm["walk_length"])).fetchone()[0]
        ref_s = time.perf_counter() - t0
        remb = np.frombuffer(b"".join(r[0] for r in c.execute("SELECT vector FROM o ORDER BY rowid")), np.float32).reshape(-1, dim)
        c.close()
        cpu = {"value": cpairs / ref_s, "unit": "pairs/s", "cores": 1, "kind": "reference", "host": host_cpu(),
               "sample": f"the reference's own node2vec_train() through SQL (oracle/_ref/muninn.so, src/node2vec.c, gcc -O2) on an ER "
                         f"graph of {cn} nodes with the same mean degree and parameters, output into a plain table: {rows} rows, "
                         f"{cpairs} pairs in {ref_s:.1f}s",
               "embedding_bytes_equal_to_port": bool(rows == cn and np.array_equal(remb.view(np.int32), cemb.view(np.int32))),
               "port": {"value": cpairs / cpu_s, "unit": "pairs/s", "cores": 1,
                        "sample": f"oracle/mn_graph_oracle.c on the same graph: {cpu_s:.1f}s"}}
    return {
        "metric": "Node2Vec (center, context) SGNS pairs/sec incl. walk generation, 1M-node / 20M-edge graph, dim 128",
        "value": pairs / wall, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"node2vec: {'Barabasi-Albert m=' + str(max(1, m // n)) if model == 'ba' else 'ER G(n,m)'} seed 42, {n} nodes, {m} edge draws -> {len(adj)} directed adjacency "
                               f"entries; p=q=1, dim {dim}, window 5, neg 5, lr 0.025, 10 walks x 80, 1 epoch; "
                               f"batch-synchronous schedule (MN_N2V_BATCHED, default batch)",
                   "nodes": n, "adjacency_entries": int(len(adj)), "pairs": int(pairs), "graph_build_s": gen_s,
                   "degree": {"mean": float(deg.mean()), "max": int(deg.max()), "p99": float(np.percentile(deg, 99))},
                   "parallelism": "single GPU" if world == 1 else
                                  f"data-parallel over {world} ranks ({args.backend}): walk/error phase split, samples all-gathered in walk "
                                  f"order, every replica applies the batch (embeddings bit-identical to 1 GPU)"},
        "parity_vs_oracle": parity,
        "to_hnsw_index": to_index,
        "embedding_norm_check": float(np.abs(np.linalg.norm(emb[:1000], axis=1) - 1.0).max()),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": _traffic("node2vec_er1M_20M_batched_default") if (n, m, world, model) == (1_000_000, 20_000_000, 1, "er") else None,
                     "per_kernel": "profiles/r03_n2v_per_kernel.json (k_n2v_walk_grad 0.48, k_n2v_apply 0.61 of 8 TB/s by their own algorithmic bytes)",
                     "kernel": "k_n2v_walk_grad + rocPRIM radix sort + k_n2v_apply (whole pipeline)",
                     "kernel_ms": dev_ms, "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": cpu,
    }


_LFR_CACHE = {}


def _nmi(a, b):
    """normalised mutual information of two partitions (pair-counting F1 is O(n^2))"""
    a = np.unique(a, return_inverse=True)[1]
    b = np.unique(b, return_inverse=True)[1]
    cont = np.zeros((a.max() + 1, b.max() + 1))
    np.add.at(cont, (a, b), 1)
    pa, pb, pab = cont.sum(1) / len(a), cont.sum(0) / len(a), cont / len(a)
    nz = pab > 0
    mi = (pab[nz] * np.log(pab[nz] / (pa[:, None] * pb[None, :])[nz])).sum()
    ha, hb = -(pa[pa > 0] * np.log(pa[pa > 0])).sum(), -(pb[pb > 0] * np.log(pb[pb > 0])).sum()
    return float(2 * mi / (ha + hb))


def bench_leiden(pkg, args):
    n = args.leiden_nodes
    quick = getattr(args, "quick", False)  # bench.py's `graph` block: bounded CPU legs
    rank, world, dist, dev = args.ctx  # run_leiden does not shard (1 GPU): N > 1 = N independent replicas
    t0 = time.perf_counter()
    if n not in _LFR_CACHE:
        _LFR_CACHE.clear()
        _LFR_CACHE[n] = pkg.lfr.lfr_like(n, 40, min(200, n // 10), 0.3)
    s, d, truth = _LFR_CACHE[n]
    # --leiden-weighted: the same graph with edge weights in [0.5, 2.5): list-order f64 sums in the evaluation and the sweep's
    # movers applied in the reference's addition order on the device (stable sort by community), instead of integer counts
    wts = (np.random.default_rng(8).random(len(s)) * 2 + 0.5) if args.leiden_weighted else None
    g = pkg.graph.graph_from_edges(n, s, d, wts, device=dev)
    gen_s = time.perf_counter() - t0
    E = len(s)
    for _ in range(args.warmup):
        g.leiden(1.0, "both", pkg.LEIDEN_BATCHED)
    walls, devs = [], []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        comm, q, st = g.leiden(1.0, "both", pkg.LEIDEN_BATCHED)
        walls.append(_max_over_ranks(dist, args, time.perf_counter() - t0))
        devs.append(st["device_ms"])
    wall, dev_ms = float(np.mean(walls)), float(np.mean(devs))
    if rank != 0:
        g.close()
        return None
    sweeps = st["move_sweeps"] + st["refine_sweeps"]
    # per sweep E_dir·(4+8+4) B (target, weight, community gather) + N·20 B; unweighted graphs carry no
    # weight array on the device, so 8 B of that is not read — kept in the figure as the survey defines it
    alg = sweeps * (2 * E * 16 + n * 20)
    achieved = alg / (dev_ms * 1e-3) / 1e9

    from oracle import orc_graph as og

    # the device's result against the CPU restatement of the same schedule (bit-exact), bounded size: the default schedule
    # (whole-graph synchronous sweeps, pick-less every 3rd = oracle batch -3) and the round schedule (rounds of 1024 nodes)
    pn = 20000
    ps, pd, _ = pkg.lfr.lfr_like(pn, 20, 100, 0.3, seed=5)
    pw = (np.random.default_rng(9).random(len(ps)) * 2 + 0.5) if args.leiden_weighted else None
    pg = pkg.graph.graph_from_edges(pn, ps, pd, pw)
    pcsr = og.Csr(ps, pd, pw, "both", n_nodes=pn, first_seen=False)
    par = {"graph": f"LFR-like n={pn}"}
    for tag, dev_batch, orc_batch in (("default_synchronous_sweeps", 0, -3), ("rounds_of_1024", 1024, 1024)):
        pc, pq, pst = pg.leiden(1.0, "both", pkg.LEIDEN_BATCHED, dev_batch)
        oc2, oq2, ost2 = og.leiden(pcsr, 1.0, orc_batch)
        par[tag] = {"communities_identical": bool(np.array_equal(pc, oc2)),
                    "modularity_bits_identical": bool(np.float64(pq).view(np.int64) == np.float64(oq2).view(np.int64)),
                    "sweeps_identical": bool((pst["move_sweeps"], pst["refine_sweeps"]) == (ost2["move_sweeps"], ost2["refine_sweeps"]))}
    pg.close()
    # CPU beside it.  (a) the reference's own compiled run_leiden (oracle/_ref, src/graph_community.c, gcc -O2) on a bounded
    # LFR-like sample of the same family; the port must return the same communities and Q bits on it.  (b) unless `quick`: the
    # port (hashed dedup instead of the reference's O(n_neigh^2) scan) on the FULL graph, for Q / NMI of the sequential schedule
    cpu = None
    if og.have_ref_graph():
        cn = getattr(args, "leiden_cpu_nodes", 150_000)
        cs, cd, _ = pkg.lfr.lfr_like(cn, 40, 200, 0.3, seed=43)
        cw = (np.random.default_rng(10).random(len(cs)) * 2 + 0.5) if args.leiden_weighted else None
        t0 = time.perf_counter()
        rc_, rq_, _ = og.ref_leiden(cs, cd, cw, "both", 1.0)
        ref_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        oc_, oq_, _ = og.leiden(og.Csr(cs, cd, cw, "both"), 1.0, 1)
        port_s = time.perf_counter() - t0
        cpu = {"value": len(cs) / ref_s, "unit": "edges/s", "cores": 1, "kind": "reference", "host": host_cpu(),
               "sample": f"the reference's own run_leiden (oracle/_ref/muninn.so, src/graph_community.c, gcc -O2) on an LFR-like graph "
                         f"of the same family, {cn} nodes / {len(cs)} edges: {ref_s:.1f}s",
               "communities_and_q_bits_equal_to_port": bool(np.array_equal(rc_, oc_) and np.float64(rq_).view(np.int64) == np.float64(oq_).view(np.int64)),
               "port": {"value": len(cs) / port_s, "unit": "edges/s", "cores": 1,
                        "sample": f"oracle/mn_graph_oracle.c (sequential schedule) on the same graph: {port_s:.1f}s"}}
    seq = None
    if not quick or cpu is None:
        csr = og.Csr(s, d, wts, "both", n_nodes=n, first_seen=False)
        t0 = time.perf_counter()
        oc, oq, ost = og.leiden(csr, 1.0, 1)  # the reference's sequential schedule
        cpu_s = time.perf_counter() - t0
        seq = {"modularity": oq, "communities": int(oc.max()) + 1, "nmi_vs_planted": _nmi(oc, truth), "seconds": cpu_s}
        if cpu is None:
            cpu = {"value": E / cpu_s, "unit": "edges/s", "cores": 1, "kind": "port", "host": host_cpu(),
                   "sample": f"the same graph, reference's sequential schedule (oracle/mn_graph_oracle.c, dedup by hashing "
                             f"instead of the reference's O(n_neigh^2) scan): {cpu_s:.1f}s"}
    tkey = f"leiden_lfr{n // 1000}k_{'weighted' if args.leiden_weighted else 'unweighted'}_sync_default"
    out = {
        "metric": "Leiden (local moving + refinement, run_leiden) input edges/sec on a 10M-edge LFR-like graph",
        "value": E * world / wall, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"leiden: LFR-like n={n}, <k>=40, k_max 200, mu=0.3, seed 42 -> {E} edges, {'weighted [0.5, 2.5)' if args.leiden_weighted else 'unweighted'}, "
                               f"direction both, resolution 1.0; MN_LEIDEN_BATCHED default schedule: whole-graph synchronous sweeps, pick-less every 3rd",
                   "nodes": n, "edges": int(E), "graph_build_s": gen_s},
        "modularity": q, "communities": int(comm.max()) + 1, "sweeps": int(sweeps), "moves": int(st["moves"]),
        "nmi_vs_planted": _nmi(comm, truth),
        "cpu_sequential": seq,
        "parity_vs_oracle": par,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": _traffic(tkey),
                     "kernel": "k_leiden_eval + k_leiden_apply_sync, one pair per sweep (whole run_leiden)", "kernel_ms": dev_ms,
                     "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": cpu,
    }
    g.close()
    return out


# ───────────────────────── SURVEY §8 f-4: graph_pagerank / graph_components / graph_node_betweenness ─────────────────────────
# Published reference numbers (benchmarks/charts/graph_query_time_{pagerank,components,betweenness}.json, series
# "muninn / erdos-renyi-20", query time through SQL): ms by node count
PUBLISHED_MS = {"pagerank": {1000: 44.423, 5000: 947.58, 10000: 3756.147, 50000: 299032.738},
                "components": {1000: 42.074, 5000: 948.109, 10000: 3747.119, 50000: 143310.545},
                "betweenness": {1000: 130.058, 5000: 3496.249, 10000: 16663.205}}
TVF_SQL = {"pagerank": "SELECT node, rank FROM graph_pagerank WHERE edge_table = 'bench_edges' AND src_col = 'src' AND dst_col = 'dst' "
                       "AND damping = 0.85 AND iterations = 100",
           "components": "SELECT node, component_id FROM graph_components WHERE edge_table = 'bench_edges' AND src_col = 'src' "
                         "AND dst_col = 'dst'",
           "betweenness": "SELECT node, centrality FROM graph_node_betweenness WHERE edge_table = 'bench_edges' AND src_col = 'src' "
                          "AND dst_col = 'dst' AND direction = 'both'"}


def er_rows(n, avg_degree, seed=42):
    """the reference harness's Erdos-Renyi model (benchmarks/harness/common.py:658-686: every pair with probability
    avg_degree / (n - 1), both directions as rows) drawn with numpy instead of n^2 / 2 calls of random.random()"""
    rng = np.random.default_rng(seed)
    m = rng.binomial(n * (n - 1) // 2, avg_degree / max(1, n - 1))
    a, b = rng.integers(0, n, int(m * 1.02) + 16), rng.integers(0, n, int(m * 1.02) + 16)
    keep = a != b
    lo, hi = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
    _, first = np.unique(lo.astype(np.int64) * n + hi, return_index=True)
    first = np.sort(first)[:m]
    lo, hi = lo[first], hi[first]
    src = np.empty(2 * len(lo), np.int32)
    dst = np.empty(2 * len(lo), np.int32)
    src[0::2], src[1::2], dst[0::2], dst[1::2] = lo, hi, hi, lo
    return src, dst


def _sql_time(so, src, dst, sql):
    """the harness's own methodology: edge table of TEXT ids, one SELECT, fetchall (graph_traversal.py:145-162)"""
    import sqlite3

    c = sqlite3.connect(":memory:")
    c.enable_load_extension(True)
    c.load_extension(so)
    c.execute("CREATE TABLE bench_edges (src TEXT, dst TEXT, weight REAL)")
    c.executemany("INSERT INTO bench_edges VALUES (?,?,1.0)", [(str(a), str(b)) for a, b in zip(src.tolist(), dst.tolist())])
    c.execute(sql).fetchall() if so.endswith("ext/muninn") else None  # (ours: one warm-up for the device's first touch)
    t0 = time.perf_counter()
    rows = c.execute(sql).fetchall()
    ms = (time.perf_counter() - t0) * 1e3
    c.close()
    return ms, rows


def bench_tvf(pkg, args, what):
    """One f-4 algorithm: (1) at the reference's published size through SQL — this extension and the compiled reference's,
    same rows, same statement, results compared; (2) at a size the device is for, through the C-ABI, with a roofline."""
    rank, world, dist, dev = args.ctx
    if rank != 0:
        return None
    ours_so = os.path.join(ROOT, "sqlite-muninn_amd", "ext", "muninn")
    ref_so = os.path.join(ROOT, "oracle", "_ref", "muninn")
    n_pub = args.tvf_published_nodes
    src, dst = er_rows(n_pub, 20)
    sql = TVF_SQL[what]
    ours_ms, ours_rows = _sql_time(ours_so, src, dst, sql)
    ref_ms, same = None, None
    if os.path.exists(ref_so + ".so") and not args.no_ref_sql:
        ref_ms, ref_rows = _sql_time(ref_so, src, dst, sql)
        if what == "components":  # ids are union-find roots (order-dependent): compare the partition
            def part(rows):
                g = {}
                for node, cid in rows:
                    g.setdefault(cid, []).append(node)
                return sorted(sorted(v) for v in g.values())
            same = part(ours_rows) == part(ref_rows)
        else:
            same = sorted(ours_rows) == sorted(ref_rows)
    # (2) the large case through the C-ABI
    if what == "betweenness":
        n_big = args.tvf_betweenness_nodes
        bs, bd = er_rows(n_big, 20)
        g = pkg.graph.graph_from_edges(n_big, bs[0::2], bd[0::2], device=dev)  # (each undirected edge once: out + in lists)
        g.betweenness("both")
        t0 = time.perf_counter()
        cb, _, dev_ms = g.betweenness("both")
        wall = time.perf_counter() - t0
        g.close()
        E = len(bs)
        # Brandes: every source walks every adjacency entry twice (BFS + accumulation): n * E_dir * 2 visits of (target 4 B +
        # sigma / dist / delta gathers 20 B)
        alg = float(n_big) * E * 2 * 24
        big = {"nodes": n_big, "edge_rows": int(E), "device_ms": dev_ms, "wall_ms": wall * 1e3,
               "sources_per_s": n_big / (dev_ms * 1e-3)}
        unit, value = "sources/s", n_big / wall
    else:
        n_big = args.tvf_nodes
        bs, bd = er_rows(n_big, 20)
        E = len(bs)
        if what == "pagerank":
            pkg.graph.pagerank(n_big, bs, bd, 0.85, 2, device=dev)
            t0 = time.perf_counter()
            _, st = pkg.graph.pagerank(n_big, bs, bd, 0.85, 100, device=dev)
            wall = time.perf_counter() - t0
            alg = 100.0 * (E * 12 + n_big * 16)  # per iteration: E (source 4 B + share 8 B) + N (rank 8 B in, 8 B out)
            big = {"nodes": n_big, "edge_rows": int(E), "iterations": 100, "device_ms": st["device_ms"], "wall_ms": wall * 1e3}
            dev_ms = st["device_ms"]
            unit, value = "edge-iterations/s", 100.0 * E / wall
        else:
            pkg.graph.components(n_big, bs, bd, pkg.graph.COMPONENTS_FAST, device=dev)
            t0 = time.perf_counter()
            _, _, st = pkg.graph.components(n_big, bs, bd, pkg.graph.COMPONENTS_FAST, device=dev)
            wall = time.perf_counter() - t0
            alg = float(st["rounds"]) * E * 16 + n_big * 12  # per hook round: E (two ends 8 B + their two parents 8 B)
            big = {"nodes": n_big, "edge_rows": int(E), "hook_rounds": st["rounds"], "device_ms": st["device_ms"], "wall_ms": wall * 1e3}
            dev_ms = st["device_ms"]
            unit, value = "edges/s", E / wall
    achieved = alg / (dev_ms * 1e-3) / 1e9
    return {
        "metric": f"graph_{'node_betweenness' if what == 'betweenness' else what} (SURVEY 8 f-4)", "value": value, "unit": unit,
        "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": big["wall_ms"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if what != "components" else "int32", "data": "synthetic",
        "config": {"workload": f"{what}: Erdos-Renyi avg degree 20 (the reference harness's model), {big['nodes']} nodes / "
                               f"{big['edge_rows']} edge rows through the C-ABI (host arrays in, result out: wall includes the "
                               f"upload and the host-side CSR build)", **big},
        "at_published_size_through_sql": {
            "graph": f"Erdos-Renyi avg degree 20, {n_pub} nodes / {len(src)} edge rows, TEXT ids, the harness's statement",
            "this_extension_ms": ours_ms, "compiled_reference_ms_on_this_host": ref_ms, "host": host_cpu(),
            "reference_published_ms": PUBLISHED_MS[what].get(n_pub), "rows_equal_to_reference": same},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": _traffic(f"tvf_{what}_er_{big['nodes']}_nodes_avg_degree_20"), "kernel_ms": dev_ms,
                     "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": None if ref_ms is None else {
            "value": ref_ms, "unit": "ms per query (lower is better)", "cores": 1, "kind": "reference", "host": host_cpu(),
            "sample": f"the compiled reference's own TVF through SQL at the published size ({n_pub} nodes): {ref_ms:.0f} ms; this "
                      f"extension on the same rows: {ours_ms:.0f} ms"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="both", choices=["node2vec", "leiden", "both", "pagerank", "components", "betweenness", "tvf"],
                    help="both = node2vec + leiden (configs 4 and 5); tvf = pagerank + components + betweenness (SURVEY 8 f-4)")
    ap.add_argument("--tvf-published-nodes", type=int, default=10_000, help="f-4: node count of the reference's published point")
    ap.add_argument("--tvf-nodes", type=int, default=1_000_000, help="f-4: pagerank / components through the C-ABI")
    ap.add_argument("--tvf-betweenness-nodes", type=int, default=20_000)
    ap.add_argument("--no-ref-sql", action="store_true", help="f-4: skip the compiled reference's SQL run")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n2v-nodes", type=int, default=1_000_000)
    ap.add_argument("--n2v-edges", type=int, default=20_000_000)
    ap.add_argument("--n2v-cpu-nodes", type=int, default=1500)
    ap.add_argument("--n2v-model", default="er", choices=["er", "ba"],
                    help="node2vec graph: er = G(n,m); ba = Barabasi-Albert with m = edges/nodes (SURVEY 8d: power-law degrees)")
    ap.add_argument("--leiden-nodes", type=int, default=500_000)
    ap.add_argument("--leiden-cpu-nodes", type=int, default=150_000, help="leiden: size of the LFR-like sample the compiled reference is timed on")
    ap.add_argument("--leiden-weighted", action="store_true", help="leiden: random edge weights (the f64 list-order path)")
    ap.add_argument("--no-index-leg", action="store_true", help="node2vec: skip the '-> hnsw index' leg (two 1M-row index builds)")
    ap.add_argument("--dump-csr", default="", help="node2vec: also write the graph as a binary CSR file (tools/n2v_bench.cpp)")
    ap.add_argument("--dump-only", action="store_true", help="with --dump-csr: write the file and stop")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1: node2vec data-parallel over N ranks (config 4); leiden = N replicas")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--device", type=int, default=-1)
    args = ap.parse_args()
    from bench import spawn_ranks_if_needed

    spawn_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    # N > 1: torch selects its device BEFORE libmuninn_hip.so touches HIP (torch carries its own HIP runtime; initialised
    # second, it reports "No HIP GPUs are available")
    args.ctx = _dist_ctx(args)
    import muninn_amd

    pkg = muninn_amd.pkg
    pkg.lib()  # fails loudly if libmuninn_hip.so is missing — there is no CPU fallback
    if pkg.device_count() < 1:
        raise SystemExit("bench_graph.py: no gfx950 device visible")
    for name, fn in (("node2vec", bench_node2vec), ("leiden", bench_leiden)):
        if args.workload in (name, "both"):
            line = fn(pkg, args)
            if line is not None:  # rank 0
                print(json.dumps(line), flush=True)
    for name in ("pagerank", "components", "betweenness"):
        if args.workload in (name, "tvf"):
            line = bench_tvf(pkg, args, name)
            if line is not None:
                print(json.dumps(line), flush=True)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
