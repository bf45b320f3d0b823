"""world_size-2 gloo tests (CPU) of the N > 1 path: the sharded-index exchange step (all-gather of
per-shard top-k + deterministic merge) checked against brute force over the union, with the CPU
oracle playing the per-shard search."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _collect(procs, q, n, timeout):
    """q.get with fail-fast: a worker that died (import error, assertion) must not cost the whole timeout"""
    import queue
    import time

    out, t0 = [], time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"worker exited with {dead}"
            assert time.time() - t0 < timeout, "timeout waiting for workers"
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muninn_amd
    from oracle import orc

    par = muninn_amd.pkg.parallel
    n, d, k, nq = 1200, 12, 10, 40
    X = np.random.default_rng(42).standard_normal((n, d), dtype=np.float32)
    ids = np.arange(1, n + 1, dtype=np.int64)
    Q = np.random.default_rng(43).standard_normal((nq, d), dtype=np.float32)
    mine = np.array([par.shard_of_rowid(i, world) == rank for i in ids])
    o = orc.Oracle(d, "l2", 8, 60)
    o.insert_many(ids[mine], X[mine])
    li, ld, lc = o.search_many(Q, k, 400)  # ef large enough that each shard search is exact
    # a duplicated vector across shards exercises the tie rule (distance, shard rank, position)
    mi, md, mc = par.allgather_merge_topk(torch.from_numpy(li), torch.from_numpy(ld), torch.from_numpy(lc), k)
    t = par.max_over_ranks(float(rank + 1), "cpu")
    q.put((rank, mi.numpy(), md.numpy(), mc.numpy(), t))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_topk_merge_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, world, 240), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n, d, k, nq = 1200, 12, 10, 40
    X = np.random.default_rng(42).standard_normal((n, d), dtype=np.float32)
    Q = np.random.default_rng(43).standard_normal((nq, d), dtype=np.float32)
    D = ((Q[:, None, :] - X[None, :, :]) ** 2).sum(2)
    truth = np.argsort(D, axis=1, kind="stable")[:, :k] + 1
    for rank, mi, md, mc, t in res:
        assert t == 2.0  # max over ranks
        assert (mc == k).all()
        assert (np.diff(md, axis=1) >= 0).all()
        assert np.mean([len(set(mi[i]) & set(truth[i])) / k for i in range(nq)]) >= 0.99
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])  # identical on every rank


def test_merge_tie_rule_single_process():
    """Equal distances: shard rank first, then position — checked without a process group by faking world=1."""
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        import muninn_amd

        par = muninn_amd.pkg.parallel
        ids = torch.tensor([[5, 7, -1]], dtype=torch.int64)
        ds = torch.tensor([[0.5, 0.5, 0.0]], dtype=torch.float32)
        cnt = torch.tensor([2], dtype=torch.int32)
        mi, md, mc = par.allgather_merge_topk(ids, ds, cnt, 3)
        assert mi.tolist() == [[5, 7, -1]] and mc.tolist() == [2]
    finally:
        dist.destroy_process_group()


def _n2v_worker(rank, world, port, q, n_nodes=1200):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muninn_amd
    from oracle import orc_graph as og
    from oracle.graph_cases import planted

    s, d, _ = planted(n_nodes, 6, 0.08, 0.002, 7)
    g = og.N2vGraph(s, d)
    emb, st = muninn_amd.pkg.parallel.node2vec_train_distributed(g.off, g.adj, 32, 1.0, 1.0, 2, 20, 3, 3, 0.025, 1, batch_walks=50)
    q.put((rank, emb, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,n_nodes", [(2, 1200), (4, 1206)])
def test_node2vec_data_parallel_world2_bit_identical_to_one_gpu(gpu, world, n_nodes):
    """Several ranks (gloo exchange, all on the one GPU of the box) train data-parallel — walk slices, samples all-gathered,
    the apply half sharded by destination row (1 206 rows over 4 ranks: a padded last shard) and the row shards all-gathered;
    every replica must equal the single-process MN_N2V_BATCHED result bit for bit, which itself equals the CPU restatement."""
    from oracle import orc_graph as og
    from oracle.graph_cases import planted

    s, d, _ = planted(n_nodes, 6, 0.08, 0.002, 7)
    g = og.N2vGraph(s, d)
    single, st1 = gpu.node2vec_train(g.off, g.adj, 32, 1.0, 1.0, 2, 20, 3, 3, 0.025, 1, mode=gpu.N2V_BATCHED, batch_walks=50)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_n2v_worker, args=(r, world, port, q, n_nodes)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, world, 300), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, emb, st in res:
        assert np.array_equal(emb.view(np.int32), single.view(np.int32)), rank
        assert st["pairs"] == st1["pairs"]


def _leiden_worker(rank, world, port, q, weighted):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muninn_amd

    pkg = muninn_amd.pkg
    n = 20_003
    s, d, _ = pkg.lfr.lfr_like(n, 20, 100, 0.3, seed=5)
    w = (np.random.default_rng(9).random(len(s)) * 2 + 0.5) if weighted else None
    g = pkg.graph.graph_from_edges(n, s, d, w)
    comm, qq, st = pkg.parallel.leiden_distributed(g, 1.0, "both", 0)
    q.put((rank, comm, qq, st))
    g.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,weighted", [(2, False), (3, True), (4, False)])
def test_leiden_divided_over_ranks_is_bit_identical_to_one_gpu(gpu, world, weighted):
    """north_star: Leiden's local-move sweep partitioned across the GPUs with the modularity partials exchanged.  Several ranks
    (gloo host transport, all on the box's one GPU) divide every synchronous sweep's evaluation and the modularity's per-node
    terms by node range (20 003 nodes: nothing divides) — every rank must return the one-GPU communities, Q bits and sweep
    counts, which are themselves the oracle's (tests/test_leiden.py)."""
    n = 20_003
    s, d, _ = gpu.lfr.lfr_like(n, 20, 100, 0.3, seed=5)
    w = (np.random.default_rng(9).random(len(s)) * 2 + 0.5) if weighted else None
    g = gpu.graph.graph_from_edges(n, s, d, w)
    c1, q1, st1 = g.leiden(1.0, "both", gpu.LEIDEN_BATCHED)
    g.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_leiden_worker, args=(r, world, port, q, weighted)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, world, 300), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, comm, qq, st in res:
        assert np.array_equal(comm, c1), rank
        assert np.float64(qq).view(np.int64) == np.float64(q1).view(np.int64), rank
        assert (st["moves"], st["move_sweeps"], st["refine_sweeps"]) == (st1["moves"], st1["move_sweeps"], st1["refine_sweeps"]), rank


def _build_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muninn_amd

    pkg = muninn_amd.pkg
    X = np.random.default_rng(77).standard_normal((6000, 24)).astype(np.float32)
    ids = np.arange(5, 6005, dtype=np.int64)
    g = pkg.HnswIndex(24, "cosine", 8, 60)
    pkg.parallel.build_distributed(g, ids, X, 16, 1024, min_split=64)
    q.put((rank, g.export_links(0), g.export_links(1), g.entry_point, g.max_level))
    g.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_build_shared_by_two_ranks_equals_one_gpu_build(gpu, world):
    """parallel.build_distributed: 2 and 4 replicas (gloo exchange, all on the box's one GPU) split the search half of
    every batch and all-gather the selected lists, and (round 4) divide the link half too — each rank replays the reverse edges
    of the targets with slot mod world == rank and the finished rows travel as records; each must end with exactly the graph
    mn_hnsw_build makes alone (layer 0 and layer 1 rows, entry point, top layer)."""
    X = np.random.default_rng(77).standard_normal((6000, 24)).astype(np.float32)
    ids = np.arange(5, 6005, dtype=np.int64)
    g = gpu.HnswIndex(24, "cosine", 8, 60)
    assert g.build(ids, X, 16, 1024) == 0
    want = (g.export_links(0), g.export_links(1), g.entry_point, g.max_level)
    g.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_build_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect(procs, q, world, 300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, l0, l1, ep, ml in res:
        assert np.array_equal(l0, want[0]) and np.array_equal(l1, want[1]), rank
        assert (ep, ml) == want[2:]


def _fault_worker(rank, world, port, q, what):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # the failure is injected on rank 1 only, in the second batch: rank 0 is healthy and must not be left inside a collective
    os.environ["MN_FAULT_INJECT"] = {"build": "build_shared:1:2000", "n2v": "n2v_shared:1:50", "search": "search_overflow:1"}[what]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muninn_amd

    pkg = muninn_amd.pkg
    msg = "no error"
    try:
        if what == "build":
            X = np.random.default_rng(77).standard_normal((6000, 24)).astype(np.float32)
            g = pkg.HnswIndex(24, "cosine", 8, 60)
            pkg.parallel.build_distributed(g, np.arange(5, 6005, dtype=np.int64), X, 16, 1024, min_split=64)
        elif what == "search":  # config 3: a shard whose heap workspace overflowed returns truncated lists
            X = np.random.default_rng(5 + rank).standard_normal((3000, 16)).astype(np.float32)
            g = pkg.HnswIndex(16, "l2", 8, 60)
            assert g.build(np.arange(3000, dtype=np.int64) * world + rank, X, 16, 1024) == 0
            comm = pkg.parallel.Comm(0)
            try:
                pkg.parallel.search_sharded(g, comm, X[:20], 5, 40)
            finally:
                comm.close()
        else:
            from oracle import orc_graph as og
            from oracle.graph_cases import planted

            s, d, _ = planted(1200, 6, 0.08, 0.002, 7)
            gr = og.N2vGraph(s, d)
            pkg.parallel.node2vec_train_distributed(gr.off, gr.adj, 32, 1.0, 1.0, 2, 20, 3, 3, 0.025, 1, batch_walks=50)
    except pkg.hnsw.MuninnHipError as e:
        msg = str(e)
    q.put((rank, msg))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_a_shard_whose_search_overflowed_fails_the_sharded_search_on_every_rank(gpu):
    """mn_hnsw_search_sharded: every shard's heap-workspace overflow count travels with its top-k lists; a shard that
    overflowed (here injected on rank 1) makes the call fail on EVERY rank, naming the shard, instead of merging a truncated
    list and returning rc 0 (the unsharded mn_hnsw_search_batch has always failed in that case)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fault_worker, args=(r, world, port, q, "search")) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(_collect(procs, q, world, 300))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        assert "shard 1" in res[r] and "exceeded heap workspace" in res[r], res


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["build", "n2v"])
def test_a_rank_local_failure_stops_every_rank_instead_of_hanging_the_job(gpu, what):
    """mn_hnsw_build_shared / mn_node2vec_train_shared: a rank whose local step fails (here injected on rank 1 in its second
    batch) still enters the status exchange, so the healthy rank is not left waiting inside the batch's all-gather: both ranks
    return an error that names the failing rank."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fault_worker, args=(r, world, port, q, what)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(_collect(procs, q, world, 300))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert "rank 1 failed" in res[0] and "all ranks stop" in res[0], res
    assert "rank 1 failed" in res[1] and "injected failure" in res[1], res


def _run_entry(script, extra, gpus=2):
    """`python <script> --gpus N ...` exactly as the driver starts it (no torchrun on the command line): the script
    itself must start its N ranks; they all share the box's one GPU (gloo exchange)."""
    import json
    import subprocess
    import sys

    cmd = [sys.executable, os.path.join(ROOT, script), "--gpus", str(gpus), "--backend", "gloo"] + extra
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return lines


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["replica", "sharded"])
def test_bench_entry_starts_its_own_ranks(gpu, mode):
    lines = _run_entry("bench.py", ["--steps", "2", "--warmup", "1", "--num-vectors", "20000", "--dim", "64", "--nq", "500",
                                    "--recall-queries", "100", "--no-cpu-baseline", "--recall-target", "0", "--mode", mode])
    assert len(lines) == 1  # rank 0 prints the one line
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0 and j["roofline"]["frac"] > 0
    assert j["config"]["parallelism"].startswith("sharded index") == (mode == "sharded")


@pytest.mark.gpu
def test_bench_graph_entry_runs_node2vec_data_parallel(gpu):
    lines = _run_entry("bench_graph.py", ["--workload", "node2vec", "--steps", "1", "--warmup", "0", "--n2v-nodes", "20000",
                                          "--n2v-edges", "200000", "--n2v-cpu-nodes", "300"])
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0
    assert lines[0]["parity_vs_oracle"]["embedding_bits_identical"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["replica", "sharded"])
def test_bench_entry_with_more_ranks_and_sizes_the_rank_count_does_not_divide(gpu, mode):
    """Rehearsal of the driver's N = 4 / 8 runs as far as one box allows (at most 6 processes may hold its GPU, this one
    included): 4 ranks, with a vector count, a query count and a batch schedule that 4 does not divide — joint build (search
    slices of unequal size, padded all-gather), query shards of unequal size, the sharded index with its 4-way merge — through
    the same entry point and the same code below the C-ABI (host transport instead of RCCL)."""
    lines = _run_entry("bench.py", ["--steps", "2", "--warmup", "1", "--num-vectors", "20003", "--dim", "48", "--nq", "501",
                                    "--recall-queries", "100", "--no-cpu-baseline", "--recall-target", "0", "--mode", mode], gpus=4)
    assert len(lines) == 1
    j = lines[0]
    assert j["n_gpus"] == 4 and j["value"] > 0 and j["roofline"]["frac"] > 0
    if mode == "replica":
        assert j["recall_at_10"] is not None and j["recall_at_10"] > 0.5  # the jointly built graph answers queries


@pytest.mark.gpu
def test_bench_graph_entry_with_four_ranks(gpu):
    """Node2Vec data-parallel over 4 ranks: walk slices, the sample all-gather and the row shards of the apply half (20 001
    rows: the last shard is padded) — embeddings of every rank equal the one-GPU bits (checked inside on a small graph)."""
    lines = _run_entry("bench_graph.py", ["--workload", "node2vec", "--steps", "1", "--warmup", "0", "--n2v-nodes", "20001",
                                          "--n2v-edges", "200000", "--n2v-cpu-nodes", "300"], gpus=4)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 4 and lines[0]["value"] > 0
    assert lines[0]["parity_vs_oracle"]["embedding_bits_identical"]
    assert abs(lines[0]["embedding_norm_check"]) < 1e-5


def test_bench_refuses_a_gpus_flag_that_disagrees_with_the_launcher():
    import subprocess
    import sys

    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr


def _cfg3_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muninn_amd

    pkg = muninn_amd.pkg
    n, d, k = 4000, 16, 10
    X = np.random.default_rng(42).standard_normal((n, d), dtype=np.float32)
    X[7] = X[6]  # the same vector in both shards: the merge's tie rule (distance, shard, position) is exercised
    ids = np.arange(n, dtype=np.int64)
    Q = np.random.default_rng(43).standard_normal((64, d), dtype=np.float32)
    Q[0] = X[6]
    mine = ids % world == rank  # config 3: shard = rowid mod world
    g = pkg.HnswIndex(d, "l2", 8, 60)
    assert g.insert_batch(ids[mine], X[mine], pkg.BUILD_SEQUENTIAL) == 0  # the reference's graph of this shard
    comm = pkg.parallel.Comm(0)
    out = [pkg.parallel.search_sharded(g, comm, Q, k, ef) for ef in (10, 80)]
    q.put((rank, out))
    comm.close()
    g.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_config3_sharded_index_on_the_hip_path_equals_the_reference_built_per_shard(gpu, orc):
    """BASELINE config 3 at a size the oracle finishes: 2 shards by rowid mod 2, each rank's HIP index searched on the
    device, per-shard top-k all-gathered and merged below the C-ABI (mn_hnsw_search_sharded).  Expected = the oracle
    (= the reference's hnsw_insert / hnsw_search, pinned) built once per shard on the same vectors + the merge in the
    total order (distance, shard, position) — SURVEY §8(e)'s oracle for this configuration.  Ids and distance bits."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cfg3_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(_collect(procs, q, world, 300), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    n, d, k = 4000, 16, 10
    X = np.random.default_rng(42).standard_normal((n, d), dtype=np.float32)
    X[7] = X[6]
    ids = np.arange(n, dtype=np.int64)
    Q = np.random.default_rng(43).standard_normal((64, d), dtype=np.float32)
    Q[0] = X[6]
    shards = []
    for r in range(world):
        o = orc.Oracle(d, "l2", 8, 60)
        assert o.insert_many(ids[ids % world == r], X[ids % world == r]) == 0
        shards.append(o)
    for j, ef in enumerate((10, 80)):
        per = [o.search_many(Q, k, ef) for o in shards]
        wi = np.full((len(Q), k), -1, np.int64)
        wd = np.zeros((len(Q), k), np.float32)
        wc = np.zeros(len(Q), np.int32)
        for qi in range(len(Q)):
            cand = [(per[r][1][qi][p], r, p, per[r][0][qi][p]) for r in range(world) for p in range(per[r][2][qi])]
            cand.sort(key=lambda t: (t[0], t[1], t[2]))
            for p, c in enumerate(cand[:k]):
                wi[qi, p], wd[qi, p] = c[3], c[0]
            wc[qi] = min(k, len(cand))
        for rank, out in res:  # every rank holds the merged result
            gi, gd, gc = out[j]
            assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.int32), wd.view(np.int32)) and np.array_equal(gc, wc), (rank, ef)
    assert set(res[0][1][0][0][0][:2].tolist()) == {6, 7}  # the duplicated vector: shard 0's copy (rowid 6) first
    assert res[0][1][0][0][0][0] == 6


@pytest.mark.gpu
def test_rccl_transport_single_rank(gpu):
    """The production transport on real hardware as far as one GPU allows: librccl is loaded on demand, a world-1
    communicator is created from a unique id, and the shared build / sharded search run their all-gathers through
    ncclAllGather on the index's stream — results equal to the plain single-GPU entry points."""
    import ctypes as C

    L = gpu.lib()
    idb = C.create_string_buffer(128)
    assert L.mn_comm_unique_id(idb) == 0, L.mn_comm_last_error()
    c = L.mn_comm_init_rccl(1, 0, idb, 0)
    assert c, L.mn_comm_last_error()
    assert L.mn_comm_world(c) == 1 and L.mn_comm_rank(c) == 0
    X = np.random.default_rng(3).standard_normal((3000, 24)).astype(np.float32)
    ids = np.arange(1, 3001, dtype=np.int64)
    Q = np.random.default_rng(4).standard_normal((50, 24)).astype(np.float32)
    a, b = gpu.HnswIndex(24, "cosine", 8, 60), gpu.HnswIndex(24, "cosine", 8, 60)
    assert a.build(ids, X, 16, 512) == 0
    assert L.mn_hnsw_build_shared(b.h, c, ids, X, len(ids), 16, 512, 64) == 0, gpu.hnsw._err()
    assert np.array_equal(a.export_links(0), b.export_links(0)) and a.entry_point == b.entry_point
    wi, wd, wc = a.search_batch(Q, 10, 64)
    gi = np.empty((50, 10), np.int64)
    gd = np.empty((50, 10), np.float32)
    gc = np.empty(50, np.int32)
    assert L.mn_hnsw_search_sharded(b.h, c, Q, 50, 10, 64, gi, gd, gc) == 0, gpu.hnsw._err()
    assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.int32), wd.view(np.int32)) and np.array_equal(gc, wc)
    a.close()
    b.close()
    L.mn_comm_destroy(c)


def _oracle_sharded(orc, shards, Q, k, ef):
    """the reference per shard + the merge in the total order (distance, shard, position)"""
    world = len(shards)
    per = [o.search_many(Q, k, ef) for o in shards]
    wi = np.full((len(Q), k), -1, np.int64)
    wd = np.zeros((len(Q), k), np.float32)
    wc = np.zeros(len(Q), np.int32)
    for qi in range(len(Q)):
        cand = [(per[r][1][qi][p], r, p, per[r][0][qi][p]) for r in range(world) for p in range(per[r][2][qi])]
        cand.sort(key=lambda t: (t[0], t[1], t[2]))
        for p, c in enumerate(cand[:k]):
            wi[qi, p], wd[qi, p] = c[3], c[0]
        wc[qi] = min(k, len(cand))
    return wi, wd, wc


@pytest.mark.gpu
@pytest.mark.parametrize("nshards", [1, 2, 3])
def test_config3_in_one_process_equals_the_reference_built_per_shard(gpu, orc, nshards):
    """mn_shards_*: the sharded index of config 3 driven by ONE host process (what a C host owning several GPUs binds).
    One box has one GPU, so the shards share ordinal 0 — streams, peer copies and the merge kernel are the ones N GPUs
    use.  Expected = the oracle (= the reference, pinned) built once per shard by exact inserts + the merge; ids and
    distance bits, before and after deletes and further inserts; negative rowids land on ((id mod n) + n) mod n."""
    n, d, k = 1500, 16, 10
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X[7] = X[6]
    ids = np.arange(n, dtype=np.int64) - 200  # negative rowids too
    Q = rng.standard_normal((48, d), dtype=np.float32)
    Q[0] = X[6]
    sh = gpu.ShardedIndex(d, "l2", 8, 60, devices=[0] * nshards)
    oracles = [orc.Oracle(d, "l2", 8, 60) for _ in range(nshards)]
    for i in range(n):  # exact inserts: hnsw_insert on the id's shard
        r = int(((ids[i] % nshards) + nshards) % nshards)
        assert gpu.lib().mn_shards_of(sh.h, int(ids[i])) == r
        assert sh.insert(ids[i], X[i]) == 0
        assert oracles[r].insert(int(ids[i]), X[i]) == 0
    for ef in (10, 80):
        gi, gd, gc = sh.search_batch(Q, k, ef)
        wi, wd, wc = _oracle_sharded(orc, oracles, Q, k, ef)
        assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.int32), wd.view(np.int32)) and np.array_equal(gc, wc), ef
    for i in rng.choice(n, 300, replace=False):
        r = int(((ids[i] % nshards) + nshards) % nshards)
        assert sh.delete(ids[i]) == 0 and oracles[r].delete(int(ids[i])) == 0
    assert sh.delete(10**9) == -1  # absent
    X2 = rng.standard_normal((200, d), dtype=np.float32)
    for j in range(200):
        nid = int(n + 1000 + j)
        assert sh.insert(nid, X2[j]) == 0 and oracles[nid % nshards].insert(nid, X2[j]) == 0
    gi, gd, gc = sh.search_batch(Q, k, 64)
    wi, wd, wc = _oracle_sharded(orc, oracles, Q, k, 64)
    assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.int32), wd.view(np.int32)) and np.array_equal(gc, wc)
    one_i, one_d = sh.search(Q[3], k, 64)  # the single-query entry (one xFilter): the same list
    assert np.array_equal(one_i, wi[3][: wc[3]]) and np.array_equal(one_d.view(np.int32), wd[3][: wc[3]].view(np.int32))
    sh.close()


@pytest.mark.gpu
def test_in_process_shards_bulk_build_searches_every_shard(gpu):
    """mn_shards_build: the shards are built side by side (one host thread per shard, batch-synchronous build) and a
    search returns exactly what each shard's own index returns, merged; near-exact recall on an easy set."""
    n, d, k, ns = 20000, 32, 10, 4
    rng = np.random.default_rng(11)
    X = rng.standard_normal((n, d), dtype=np.float32)
    ids = np.arange(1, n + 1, dtype=np.int64)
    Q = X[rng.choice(n, 64, replace=False)] + 0.01 * rng.standard_normal((64, d), dtype=np.float32)
    sh = gpu.ShardedIndex(d, "l2", 16, 100, devices=[0] * ns)
    assert sh.build(ids, X) == 0, sh._err()
    gi, gd, gc = sh.search_batch(Q, k, 200)
    # the same shards built one by one with the single-index entry point give the same lists → same merged result
    per = []
    for r in range(ns):
        g = gpu.HnswIndex(d, "l2", 16, 100)
        m = ids % ns == r
        assert g.build(ids[m], X[m]) == 0
        per.append(g.search_batch(Q, k, 200))
        g.close()
    for qi in range(len(Q)):
        cand = [(per[r][1][qi][p], r, p, per[r][0][qi][p]) for r in range(ns) for p in range(per[r][2][qi])]
        cand.sort(key=lambda t: (t[0], t[1], t[2]))
        assert [c[3] for c in cand[:k]] == gi[qi].tolist()
        assert np.array_equal(np.array([c[0] for c in cand[:k]], np.float32).view(np.int32), gd[qi].view(np.int32))
    D = ((Q[:, None, :] - X[None, :, :]) ** 2).sum(2)
    truth = np.argsort(D, axis=1, kind="stable")[:, :k] + 1
    assert np.mean([len(set(gi[i]) & set(truth[i])) / k for i in range(len(Q))]) >= 0.97
    assert (gc == k).all()
    sh.close()


_NCCL_COEXIST = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))   # torch's own RCCL communicator first, as bench.py does
t = torch.ones(4, device="cuda"); dist.all_reduce(t)                 # ... and used
sys.path.insert(0, os.environ["MN_ROOT"])
import muninn_amd
pkg = muninn_amd.pkg
comm = pkg.parallel.Comm(0)                                          # mn_comm: id via broadcast_object_list, ncclCommInitRank
n, d = 6000, 24
X = np.random.default_rng(1).standard_normal((n, d), dtype=np.float32)
ids = np.arange(1, n + 1, dtype=np.int64)
Q = np.random.default_rng(2).standard_normal((50, d), dtype=np.float32)
a = pkg.HnswIndex(d, "cosine", 8, 60); b = pkg.HnswIndex(d, "cosine", 8, 60)
pkg.parallel.build_distributed(a, ids, X, 16, 1024, comm=comm)       # all-gathers through ncclAllGather on the index's stream
assert b.build(ids, X, 16, 1024) == 0
assert a.graph(ids) == b.graph(ids)
si, sd, sc = pkg.parallel.search_sharded(a, comm, Q, 10, 64)
pi, pd, pc = b.search_batch(Q, 10, 64)
assert np.array_equal(si, pi) and np.array_equal(sd.view(np.int32), pd.view(np.int32)) and np.array_equal(sc, pc)
off = np.arange(0, 2 * 400 + 1, 2, dtype=np.int32); adj = np.stack([(np.arange(400) + 1) % 400, (np.arange(400) - 1) % 400], 1).astype(np.int32).ravel()
e1, st1 = pkg.parallel.node2vec_train_distributed(off, adj, 16, num_walks=2, walk_length=10, device=0, comm=comm)
e2, st2 = pkg.graph.node2vec_train(off, adj, 16, num_walks=2, walk_length=10, mode=pkg.N2V_BATCHED)
assert np.array_equal(e1.view(np.int32), e2.view(np.int32))
t2 = torch.ones(4, device="cuda"); dist.all_reduce(t2)               # torch's communicator still works afterwards
comm.close(); dist.destroy_process_group()
print("NCCL_COEXIST_OK")
"""


@pytest.mark.gpu
def test_mn_comm_next_to_torchs_own_rccl_communicator(gpu):
    """bench.py's N > 1 flow inside one process of world 1 (all one GPU allows): torch.distributed's RCCL process group is
    created and used first, then mn_comm creates its own communicator from an id sent with broadcast_object_list, and
    the shared build, the sharded search and the data-parallel Node2Vec run their all-gathers through it — same results
    as the single-GPU entry points, and torch's communicator is still usable afterwards."""
    import subprocess
    import sys

    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               MN_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_COEXIST], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_COEXIST_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_in_process_shards_with_empty_and_short_shards(gpu):
    """fewer nodes than shards / than k: empty shards contribute nothing, the merged list is as long as the index"""
    d, k = 8, 5
    rng = np.random.default_rng(3)
    sh = gpu.ShardedIndex(d, "l2", 4, 20, devices=[0, 0, 0])
    Q = rng.standard_normal((3, d), dtype=np.float32)
    gi, gd, gc = sh.search_batch(Q, k, 10)  # nothing inserted yet
    assert (gc == 0).all() and (gi == -1).all()
    X = rng.standard_normal((2, d), dtype=np.float32)
    assert sh.insert(3, X[0]) == 0 and sh.insert(7, X[1]) == 0  # shards 0 and 1; shard 2 stays empty
    assert sh.insert(3, X[0]) == -1  # duplicate rowid, as hnsw_insert
    gi, gd, gc = sh.search_batch(Q, k, 10)
    assert (gc == 2).all() and (gi[:, 2:] == -1).all()
    for qi in range(3):
        want = sorted([(float(((Q[qi] - X[0]) ** 2).sum()), 3), (float(((Q[qi] - X[1]) ** 2).sum()), 7)])
        assert gi[qi, :2].tolist() == [w[1] for w in want]
        assert (np.diff(gd[qi, :2]) >= 0).all()
    ids1, _ = sh.search(Q[0], k, 10)
    assert ids1.tolist() == gi[0, :2].tolist()
    sh.close()
