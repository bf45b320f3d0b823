#!/usr/bin/env python3
"""bench.py — sqlite-muninn hot path on MI355X: batched kNN over a device-resident HNSW index.

Workload (BASELINE.json configs[1]): N x 768 f32 vectors (default N = 1M), index built on the
GPU (batch-synchronous schedule, M=16, ef_construction=200, cosine), then one STEP = one
10k-query batched kNN (k=10, ef=128) with queries, index and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: queries are independent — every rank holds a replica of the index (same
seed → identical graph) and searches its own 10k-query batch; no data-path collective;
value = all ranks' queries / max-over-ranks time ("weak").

Rank 0 prints ONE JSON line (metric/value/... + roofline + cpu_baseline).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0  # same guide: what a float4 copy kernel reaches on this part (BASELINE.md §3 reports both)


def spawn_ranks_if_needed(gpus: int, script: str, argv: list) -> None:
    """`python bench.py --gpus N` (no torchrun on the command line) must still run N ranks: when N > 1 and this process
    is not already a rank, start `python -m torch.distributed.run --nproc-per-node N <script> <argv>` as a CHILD and
    exit with its code.  Called before torch / HIP are touched (a process that has initialised the GPU is never
    re-executed); rank 0 of the children prints the one JSON line, which is relayed as is."""
    in_rank = "WORLD_SIZE" in os.environ
    if in_rank:
        if int(os.environ["WORLD_SIZE"]) != gpus:
            raise SystemExit(f"{os.path.basename(script)}: --gpus {gpus} disagrees with WORLD_SIZE={os.environ['WORLD_SIZE']}")
        return
    if gpus <= 1:
        return
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def dataset_spec(dataset):
    """("gaussian" | "lowrank" | "clustered", parameter): "lowrank16" = intrinsic dimension 16 (default 32);
    "clustered0.01" = within-cluster sigma 0.01 (default 0.1)."""
    for kind in ("lowrank", "clustered", "gaussian"):
        if dataset.startswith(kind):
            rest = dataset[len(kind):]
            return kind, (float(rest) if rest else {"lowrank": 32.0, "clustered": 0.1, "gaussian": 0.0}[kind])
    raise SystemExit(f"bench.py: unknown dataset {dataset!r}")


def dataset_note(dataset, dim):
    """what the data is, with an intrinsic dimension a reader can hold the recall against"""
    kind, par = dataset_spec(dataset)
    if kind == "gaussian":
        return f"isotropic standard normal, intrinsic dimension {dim} (the worst case for any graph index)"
    if kind == "lowrank":
        return f"x = zA + 0.02 eps, z in R^{int(par)}: intrinsic dimension {int(par)} + 2% isotropic noise, unit-normalised"
    # 64 unit centres (nearly orthogonal in 768-d) + sigma*eps: covariance eigenvalues 64 x (1/64 + s^2), (dim-64) x s^2
    s2 = par * par
    pr = (1 + dim * s2) ** 2 / (64 * (1 / 64 + s2) ** 2 + (dim - 64) * s2 * s2)
    return (f"64 Gaussian clusters, sigma {par:g}, unit-normalised (SURVEY 8d): participation-ratio dimension {pr:.0f}; inside a "
            f"cluster the data is isotropic in all {dim} dimensions")


def _gen_chunk(kind, par, dim, seed, c, m, shared):
    """chunk c of a derived dataset: its own generator seeded by (seed, c), so that chunks can be drawn side by side"""
    rng = np.random.default_rng([seed, c])
    if kind == "lowrank":
        r = int(par)
        v = rng.standard_normal((m, r), dtype=np.float32) @ shared
        v += np.float32(0.02) * rng.standard_normal((m, dim), dtype=np.float32)
    else:
        v = np.float32(par) * rng.standard_normal((m, dim), dtype=np.float32)
        v += shared[rng.integers(0, 64, m)]
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v


def gen_chunks(n, dim, seed, dataset, chunk=65536):
    """Seeded synthetic vectors as a stream of 64k-row chunks — the same rows whatever is done with them.
    gaussian: ONE default_rng(seed) stream, as the survey defines it.  The derived sets (lowrank*, clustered*) draw every chunk
    from default_rng([seed, chunk index]), which lets a few host threads generate a million rows in seconds."""
    kind, par = dataset_spec(dataset)
    if kind == "gaussian":
        rng = np.random.default_rng(seed)
        for a in range(0, n, chunk):
            yield rng.standard_normal((min(n, a + chunk) - a, dim), dtype=np.float32)
        return
    if kind == "lowrank":
        # embedding-like data: intrinsic dimension r embedded in `dim` (x = zA + 0.02 eps), unit-normalised.  Isotropic Gaussian
        # and the sigma = 0.1 "clustered" set are both ~768-dimensional intrinsically (the cluster noise 0.1*sqrt(768) dwarfs
        # the unit centres), where every graph index has poor recall.
        r = int(par)
        shared = np.random.default_rng(777).standard_normal((r, dim), dtype=np.float32) / np.float32(np.sqrt(r))
    else:  # "clustered": 64 Gaussian clusters, sigma par, unit-normalised; centres shared by base vectors and queries
        shared = np.random.default_rng(4242).standard_normal((64, dim), dtype=np.float32)
        shared /= np.linalg.norm(shared, axis=1, keepdims=True)
    from concurrent.futures import ThreadPoolExecutor

    starts = list(range(0, n, chunk))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:  # (numpy's generators release the GIL)
        for v in ex.map(lambda a: _gen_chunk(kind, par, dim, seed, a // chunk, min(n, a + chunk) - a, shared), starts):
            yield v


def gen_vectors(n, dim, seed, dataset, chunk=65536):
    out = np.empty((n, dim), np.float32)
    a = 0
    for v in gen_chunks(n, dim, seed, dataset, chunk):
        out[a:a + len(v)] = v
        a += len(v)
    return out


def build_streamed(g, n, dim, seed, dataset, first_id=1, piece=1 << 20):
    """Build without ever holding the whole matrix on the host (10M x 768 = 30.7 GB): the chunk stream is fed to
    mn_hnsw_build a million rows at a time — the same vectors and ids as one call over the whole matrix; the batch schedule
    differs only at the piece boundaries (the last batch of a piece is cut short)."""
    buf, have, done, spent = [], 0, 0, 0.0
    for v in gen_chunks(n, dim, seed, dataset):
        buf.append(v)
        have += len(v)
        if have >= piece or done + have == n:
            X = np.concatenate(buf)
            ids = np.arange(first_id + done, first_id + done + have, dtype=np.int64)
            t0 = time.perf_counter()
            if g.build(ids, X, 16, 8192) != 0:
                return -1, spent
            g.sync()
            spent += time.perf_counter() - t0  # (generating the rows is not part of the build)
            done += have
            buf, have = [], 0
            progress(f"  built {done} / {n}")
    return 0, spent


def progress(msg):
    """one line per stage on stderr (stdout carries only the JSON line): long runs must not look hung"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def recall_of(found, truth, k):
    return float(np.mean([len(set(found[i].tolist()) & set(truth[i].tolist())) / k for i in range(len(truth))]))


def graph_quality_leg(pkg, args, dev_ord, order, M, EFC, n, datasets, efs):
    """Is the graph of the batch-synchronous build as good as the reference's one-at-a-time graph?  Same vectors, same
    parameters, two indexes: MN_BUILD_SEQUENTIAL (bit-identical to the reference's hnsw_insert loop — tests/golden) and
    the batched schedule bench.py's headline index is built with; recall@k of both against exact ground truth
    (k_brute_mfma) over ALL queries, at each ef."""
    D, NQ, K = args.dim, args.nq, args.k
    out = []
    for ds in datasets:
        X = gen_vectors(n, D, 42, ds)
        Q = gen_vectors(NQ, D, 43, ds)
        ids = np.arange(1, n + 1, dtype=np.int64)
        row = {"dataset": ds, "n": n, "dim": D, "queries": NQ, "k": K}
        truth = None
        for name, mode in (("exact", "sequential"), ("batched", "batched")):
            g = pkg.HnswIndex(D, args.metric, M, EFC, order=order, device=dev_ord)
            progress(f"graph quality: {ds} {n}x{D}, {name} build")
            t0 = time.perf_counter()
            if mode == "sequential":  # in pieces, so that a multi-minute exact build reports progress
                rc, step = 0, 20_000
                for a in range(0, n, step):
                    rc = rc or g.insert_batch(ids[a:a + step], X[a:a + step], pkg.BUILD_SEQUENTIAL)
                    progress(f"  exact inserts: {min(n, a + step)} / {n}")
            else:
                rc = g.build(ids, X, 16, 8192)
            if rc != 0:
                raise SystemExit("build failed: " + pkg.hnsw._err())
            g.sync()
            row[f"{name}_build_vectors_per_s"] = n / (time.perf_counter() - t0)
            dq = g.dev_malloc(Q.nbytes)
            g.dev_upload(dq, Q)
            if truth is None:
                truth = g.bruteforce_topk(dq, NQ, K)
            d_ids, d_ds, d_cnt = g.dev_malloc(NQ * K * 8), g.dev_malloc(NQ * K * 4), g.dev_malloc(NQ * 4)
            for ef in efs:
                g.search_batch_dev(dq, NQ, K, ef, d_ids, d_ds, d_cnt)
                g.sync()
                o = np.empty((NQ, K), np.int64)
                g.dev_download(o, d_ids)
                row[f"{name}_recall_ef{ef}"] = recall_of(o, truth, K)
            g.close()
        row["max_abs_recall_gap"] = max(abs(row[f"exact_recall_ef{ef}"] - row[f"batched_recall_ef{ef}"]) for ef in efs)
        out.append(row)
    return out


def wave_order_leg(pkg, args, dev_ord, M, EFC, X, Q):
    """The headline is measured in the reference's summation order (bits identical to the reference).  north_star's bar for
    floating point is looser — bit-exact neighbour-id SETS, distances within 1e-5 relative — and the wavefront-native order
    (coalesced float4 per lane + butterfly) meets it at a higher fraction of the roofline: same data, same parameters, index
    built and searched in MN_ORDER_WAVE, results held against the compiled reference's hnsw_search on the same graph."""
    from oracle import orc

    N, D, NQ, K, EF = args.n, args.dim, args.nq, args.k, args.ef
    g = pkg.HnswIndex(D, args.metric, M, EFC, order=pkg.ORDER_WAVE, device=dev_ord)
    if g.build(np.arange(1, N + 1, dtype=np.int64), X, 16, 8192) != 0:
        raise SystemExit("build failed: " + pkg.hnsw._err())
    dq = g.dev_malloc(Q.nbytes)
    g.dev_upload(dq, Q)
    d_ids, d_ds, d_cnt = g.dev_malloc(NQ * K * 8), g.dev_malloc(NQ * K * 4), g.dev_malloc(NQ * 4)
    for _ in range(2):
        g.search_batch_dev(dq, NQ, K, EF, d_ids, d_ds, d_cnt)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g.search_batch_dev(dq, NQ, K, EF, d_ids, d_ds, d_cnt)
    g.sync()
    wall = (time.perf_counter() - t0) / args.steps
    kms = []
    for _ in range(5):
        g.search_batch_dev(dq, NQ, K, EF, d_ids, d_ds, d_cnt)
        st = g.last_launch()
        kms.append(st["last_kernel_ms"])
    alg = st["last_n_dist"] * D * 4 + st["last_n_expanded"] * (2 * M) * 4 + st["last_n_dist"] * 4
    out = {"order": "MN_ORDER_WAVE", "queries_per_s": NQ / wall, "kernel_ms": float(np.mean(kms)),
           "roofline_frac": alg / (np.mean(kms) * 1e-3) / 1e9 / HBM_PEAK_GBS, "n_dist_per_query": st["last_n_dist"] / NQ}
    gi = np.empty((NQ, K), np.int64)
    gd = np.empty((NQ, K), np.float32)
    g.dev_download(gi, d_ids)
    g.dev_download(gd, d_ds)
    nr = min(args.ref_queries, NQ)
    if orc.have_ref() and nr > 0:  # the reference's own compiled search (its summation order) on this graph
        r = orc.Ref(D, args.metric, M, EFC)
        r.load_from_device(g, vectors=X)
        ri, rd, _ = r.search_many(Q[:nr], K, EF)
        out["vs_reference_binary"] = {
            "queries": nr, "id_sets_identical": int(sum(set(ri[i].tolist()) == set(gi[i].tolist()) for i in range(nr))),
            "max_rel_distance_diff": float((np.abs(rd - gd[:nr]) / np.maximum(np.abs(rd), 1.0)).max())}
        del r
    else:
        o = orc.Oracle(D, args.metric, M, EFC, order=orc.ORDER_SSE)
        o.load_from_device(g, vectors=X)
        nr = min(500, NQ)
        ri, rd, _ = o.search_many(Q[:nr], K, EF)
        out["vs_reference_order"] = {
            "queries": nr, "id_sets_identical": int(sum(set(ri[i].tolist()) == set(gi[i].tolist()) for i in range(nr))),
            "max_rel_distance_diff": float((np.abs(rd - gd[:nr]) / np.maximum(np.abs(rd), 1.0)).max())}
    g.close()
    return out


def recall_target_leg(pkg, args, dev_ord, order, M, EFC, target, datasets):
    """north_star's target reads "kNN queries/s at recall@10 >= 0.95": isotropic 768-d Gaussian data cannot reach that
    with the reference's algorithm at any practical ef.  The same measurement — same index parameters, same
    kernel, same batch size, recall against device brute force over all queries — is therefore repeated on data whose
    intrinsic dimension is STATED (embedding-like sets of intrinsic dimension 16 / 32 / 64, and SURVEY §8d's clustered set at
    two sigmas), each with the smallest ef of a fixed ladder that reaches the target, or "not reached" up to ef 1024."""
    N, D, NQ, K = args.n, args.dim, args.nq, args.k
    rows = []
    for ds in datasets:
        progress(f"recall-target leg: {ds}")
        X = gen_vectors(N, D, 42, ds)
        Q = gen_vectors(NQ, D, 43, ds)
        g = pkg.HnswIndex(D, args.metric, M, EFC, order=order, device=dev_ord)
        t0 = time.perf_counter()
        if g.build(np.arange(1, N + 1, dtype=np.int64), X, 16, 8192) != 0:
            raise SystemExit("build failed: " + pkg.hnsw._err())
        g.sync()
        build_s = time.perf_counter() - t0
        del X
        dq = g.dev_malloc(Q.nbytes)
        g.dev_upload(dq, Q)
        d_ids, d_ds, d_cnt = g.dev_malloc(NQ * K * 8), g.dev_malloc(NQ * K * 4), g.dev_malloc(NQ * 4)
        nrec = min(max(args.recall_queries, 1), NQ)
        truth = g.bruteforce_topk(dq, nrec, K)
        ladder, hit = [], None
        for ef in (64, 128, 256, 512, 1024):
            if ef < K:
                continue
            g.search_batch_dev(dq, NQ, K, ef, d_ids, d_ds, d_cnt)  # warm-up
            g.sync()
            kms = []
            t0 = time.perf_counter()
            for _ in range(3):  # (per-launch HIP-event time and the wall time of the same three launches)
                g.search_batch_dev(dq, NQ, K, ef, d_ids, d_ds, d_cnt)
                st = g.last_launch()
                kms.append(st["last_kernel_ms"])
            wall = (time.perf_counter() - t0) / 3
            out = np.empty((NQ, K), np.int64)
            g.dev_download(out, d_ids)
            rec = recall_of(out[:nrec], truth, K)
            alg = st["last_n_dist"] * D * 4 + st["last_n_expanded"] * (2 * M) * 4 + st["last_n_dist"] * 4
            row = {"ef": ef, "queries_per_s": NQ / wall, "recall_at_10": rec, "n_dist_per_query": st["last_n_dist"] / NQ,
                   "kernel_ms": float(np.mean(kms)), "roofline_frac": alg / (np.mean(kms) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            ladder.append(row)
            if rec >= target:
                hit = row
                break
        g.close()
        rows.append({"dataset": ds, "what": dataset_note(ds, D), "build_vectors_per_s": N / build_s,
                     "reached": hit if hit else f"not reached up to ef {ladder[-1]['ef']} (recall {ladder[-1]['recall_at_10']:.3f})",
                     "ladder": ladder})
    return {"target_recall_at_10": target,
            "setup": f"{N}x{D} f32, {NQ} queries, k={K}, {args.metric}, M={M} efC={EFC}; recall over {min(max(args.recall_queries, 1), NQ)} "
                     f"queries against k_brute_mfma", "datasets": rows}


def graph_block(pkg, args, dev_ord, t_start):
    """BASELINE configs 4 and 5 in the driver's own line (every graph-half figure used to be a builder claim under
    profiles/): one run_leiden on config 5's graph, unweighted and weighted, and one Node2Vec training run on config 4's graph
    with its "-> hnsw index" leg — each the line bench_graph.py prints for that workload (value, kernel_ms, roofline with PMC
    traffic when the kernels are the profiled ones, cpu_baseline from the compiled reference, parity against the oracle).  A leg
    is skipped, with the reason, when the run would pass --graph-budget-s (the whole default run has to stay within minutes)."""
    import bench_graph as bg

    def ns(**kw):
        base = dict(steps=2, warmup=1, n2v_nodes=1_000_000, n2v_edges=20_000_000, n2v_cpu_nodes=1500, n2v_model="er",
                    leiden_nodes=500_000, leiden_cpu_nodes=150_000, leiden_weighted=False, no_index_leg=False, dump_csr="",
                    dump_only=False, gpus=1, backend=args.backend, device=dev_ord, ctx=(0, 1, None, dev_ord), quick=True)
        base.update(kw)
        return argparse.Namespace(**base)

    out = {}
    legs = (("leiden_config5", 40, lambda: bg.bench_leiden(pkg, ns())),
            ("leiden_config5_weighted", 35, lambda: bg.bench_leiden(pkg, ns(leiden_weighted=True))),
            ("node2vec_config4", 90, lambda: bg.bench_node2vec(pkg, ns(steps=1, warmup=0))))
    for name, est_s, fn in legs:
        spent = time.perf_counter() - t_start
        if spent + est_s > args.graph_budget_s:
            out[name] = {"skipped": f"{spent:.0f}s of the run already spent, this leg needs about {est_s}s, --graph-budget-s {args.graph_budget_s}"}
            continue
        progress(f"graph block: {name}")
        t0 = time.perf_counter()
        line = fn()
        line["leg_wall_s"] = time.perf_counter() - t0
        out[name] = line
    return out


def host_cpu():
    """model name and core counts of the host the CPU baselines ran on (north_star: "count stated")"""
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        vis = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        vis = os.cpu_count()
    return {"model": model, "logical_cpus": os.cpu_count(), "visible_to_this_process": vis}


def kernel_sources_sha(files):
    """sha-256 over the sources a kernel is compiled from: profiles/traffic.json stamps a PMC measurement with it, and a
    measurement whose kernel has changed since is not reported (it would silently go stale otherwise)"""
    import hashlib

    h = hashlib.sha256()
    for f in files:
        try:
            h.update(open(os.path.join(ROOT, "sqlite-muninn_amd", "csrc", f), "rb").read())
        except OSError:
            return None
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num-vectors", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--order", default="sse", choices=["sse", "wave"])
    ap.add_argument("--dataset", default="gaussian",
                    help="gaussian | lowrank[R] (intrinsic dimension R, default 32) | clustered[SIGMA] (default 0.1)")
    ap.add_argument("--recall-queries", type=int, default=-1,
                    help="queries whose exact top-k (k_brute_mfma, the MFMA GEMM + fused top-k) recall is measured against; -1 = all")
    ap.add_argument("--quality-n", type=int, default=50_000,
                    help="N=1: size of the exact-vs-batched graph comparison (two extra builds; 0 = skip)")
    ap.add_argument("--quality-datasets", default="lowrank", help="comma list of datasets for that comparison")
    ap.add_argument("--stream-above", type=int, default=2_000_000,
                    help="N above this is built from the chunk stream without holding the matrix on the host (single rank)")
    ap.add_argument("--exact-inserts", type=int, default=600,
                    help="N=1: vectors inserted one at a time (reference semantics) into the full-size index, by the GPU and by "
                         "the compiled reference on the same graph: build CPU baseline + full-size insert parity (0 = skip)")
    ap.add_argument("--cpu-queries", type=int, default=4000)
    ap.add_argument("--ref-queries", type=int, default=1500,
                    help="queries timed through the compiled reference (oracle/_ref) when it is present")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wave-leg", action="store_true",
                    help="skip the extra leg that repeats the measurement in the wavefront-native summation order")
    ap.add_argument("--recall-target", type=float, default=0.95,
                    help="N=1, gaussian only: also report q/s at the smallest ef reaching this recall@10 on "
                         "embedding-like data (0 = skip)")
    ap.add_argument("--ef-sweep", default="auto",
                    help="comma list of ef values reported next to the headline (q/s, recall, distances per query) on the same "
                         "index; auto = 128,256,512 at N=1 and nothing for N>1; '' = none")
    ap.add_argument("--target-datasets", default="lowrank16,lowrank32,lowrank64,clustered,clustered0.01",
                    help="datasets of the recall-target leg (each builds its own full-size index)")
    ap.add_argument("--no-graph-block", action="store_true",
                    help="N=1: skip the `graph` block (configs 4 and 5: Leiden unweighted + weighted, Node2Vec -> index)")
    ap.add_argument("--graph-budget-s", type=float, default=480.0,
                    help="a leg of the graph block is skipped when the run's wall time so far plus the leg's estimate exceeds this")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--mode", default="replica", choices=["replica", "sharded"],
                    help="N>1: replica = same index on every GPU, queries sharded (no collective); sharded = config 3: "
                         "rowid mod N shards, same queries everywhere, RCCL all-gather + merge of per-shard top-k")
    args = ap.parse_args()
    t_run0 = time.perf_counter()
    dataset_spec(args.dataset)
    spawn_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        if args.device < 0 and args.backend == "gloo":  # rehearsal: more ranks than GPUs share the cards round-robin
            args.device = local_rank % max(1, torch.cuda.device_count())
        dev_ord = args.device if args.device >= 0 else local_rank
        torch.cuda.set_device(dev_ord)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_ord))
        else:
            dist.init_process_group("gloo")

    import muninn_amd

    pkg = muninn_amd.pkg
    pkg.lib()  # fails loudly if libmuninn_hip.so is missing — there is no CPU fallback
    if pkg.device_count() < 1:
        raise SystemExit("bench.py: no gfx950 device visible")

    N, D, NQ, K, EF = args.n, args.dim, args.nq, args.k, args.ef
    if args.recall_queries < 0:
        args.recall_queries = NQ
    order = pkg.ORDER_SSE if args.order == "sse" else pkg.ORDER_WAVE
    M, EFC = 16, 200

    # ---- synthetic data (same seed on every rank → replicas are identical) ----
    sharded = args.mode == "sharded" and world > 1
    streamed = N > args.stream_above  # e.g. 10M x 768: 30.7 GB fits HBM (288 GB) but is not held on the host
    if streamed:
        if world > 1 or not args.no_cpu_baseline or args.recall_target > 0 or args.quality_n > 0:
            progress("large N: streamed build; CPU baselines, the recall-target leg and the graph-quality leg are skipped")
        args.no_cpu_baseline, args.recall_target, args.quality_n = True, 0.0, 0
        if world > 1:
            raise SystemExit("bench.py: the streamed build is single-rank")
        X, ids = None, None
        Q = gen_vectors(NQ, D, 43 + rank, args.dataset)
    elif sharded:  # shard r holds the rowids ≡ r (mod world); every rank searches the SAME queries
        X = gen_vectors(N, D, 42 + 1000 * rank, args.dataset)
        ids = np.arange(N, dtype=np.int64) * world + rank
        Q = gen_vectors(NQ, D, 43, args.dataset)
    else:
        X = gen_vectors(N, D, 42, args.dataset)
        ids = np.arange(1, N + 1, dtype=np.int64)
        Q = gen_vectors(NQ, D, 43 + rank, args.dataset)

    progress(f"data generated; building {N}x{D} on device {args.device if args.device >= 0 else local_rank}")
    # ---- build on the device (reported, not the timed step) ----
    dev_ord = args.device if args.device >= 0 else local_rank
    g = pkg.HnswIndex(D, args.metric, M, EFC, order=order, device=dev_ord)
    shared_build = world > 1 and not sharded and os.environ.get("MN_BENCH_SHARED_BUILD", "1") != "0"
    comm = None
    if dist is not None:
        # the exchange itself runs below the C-ABI (mn_comm: RCCL all-gather on the library's stream, or the host transport
        # of the gloo rehearsal); torch.distributed only hands rank 0's RCCL id to the others and provides barriers
        comm_err = ""
        try:
            comm = pkg.parallel.Comm(dev_ord)
        except Exception as e:  # e.g. librccl cannot be loaded next to torch's: every rank must take the same branch
            comm, comm_err = None, repr(e)[:200]
        import torch

        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            progress(f"mn_comm unavailable on some rank ({comm_err or 'another rank'}): every rank builds its replica alone "
                     f"(same seed, same graph); the timed search step has no exchange either way")
            if comm is not None:
                comm.close()
            comm, shared_build = None, False
            if sharded:
                raise SystemExit("bench.py --mode sharded needs the exchange: " + comm_err)
        dist.barrier()
    t0 = time.perf_counter()
    streamed_build_s = None
    if streamed:
        rc, streamed_build_s = build_streamed(g, N, D, 42, args.dataset)
        if rc != 0:
            raise SystemExit("build failed: " + pkg.hnsw._err())
    elif shared_build:
        # replicas of ONE graph: every batch's search half is split over the ranks, the selected lists are all-gathered
        # (RCCL), every replica links the whole batch → the graph of a one-GPU build, on every GPU (parallel.py)
        pkg.parallel.build_distributed(g, ids, X, 16, 8192, comm=comm)
    elif g.build(ids, X, 16, 8192) != 0:
        raise SystemExit("build failed: " + pkg.hnsw._err())
    g.sync()
    build_s = streamed_build_s if streamed else time.perf_counter() - t0
    build_s_max = build_s
    bst = g.build_stats()
    if dist is not None:  # slowest rank's build: the N-GPU build rate is (vectors built by all ranks) / that
        import torch

        tb = torch.tensor([build_s], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        build_s_max = float(tb.item())

    # ---- HBM-resident inputs / outputs ----
    dq = g.dev_malloc(Q.nbytes)
    g.dev_upload(dq, Q)
    d_ids = g.dev_malloc(NQ * K * 8)
    d_ds = g.dev_malloc(NQ * K * 4)
    d_cnt = g.dev_malloc(NQ * 4)

    def barrier():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()
        g.sync()

    def run_steps(nsteps, ef, collect=False, merge=True):
        kms, nd, ne = [], 0, 0
        for _ in range(nsteps):
            if sharded and merge:  # shard search + the one exchange step (all-gather of per-shard top-k) + device merge
                pkg.parallel.search_sharded_dev(g, comm, dq, NQ, K, ef, d_ids, d_ds, d_cnt)
            else:
                g.search_batch_dev(dq, NQ, K, ef, d_ids, d_ds, d_cnt)
            if collect:  # per-launch HIP-event time on the kernel's own stream (syncs that launch)
                st = g.last_launch()
                kms.append(st["last_kernel_ms"])
                nd, ne = st["last_n_dist"], st["last_n_expanded"]
                if st["last_n_overflow"]:
                    raise SystemExit("heap workspace overflow — results would be invalid")
        return kms, nd, ne

    progress(f"built in {build_s:.1f}s; timing {args.steps} steps")
    run_steps(args.warmup, EF)
    barrier()
    t0 = time.perf_counter()
    # K launches on the index's stream; every launch is followed by the read-back of its HIP events and device counters (a
    # few bytes; its synchronisation is inside the timed region), so that kernel_ms and ms_per_step come from the SAME launches
    kms, n_dist, n_exp = run_steps(args.steps, EF, collect=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel time + algorithmic bytes, measured live with HIP events over the timed launches ----
    kernel_ms = float(np.mean(kms))
    # bytes/query = n_dist*dim*4 (candidate rows) + n_expanded*2M*4 (link rows) + n_dist*4 (visited)
    alg_bytes = n_dist * D * 4 + n_exp * (2 * M) * 4 + n_dist * 4
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    # ---- recall@k against exact brute force on the device ----
    out_ids = np.empty((NQ, K), np.int64)
    g.dev_download(out_ids, d_ids)
    nrec = min(args.recall_queries, NQ)
    recall = None
    if sharded:
        nrec = 0  # (ground truth lives on the union of the shards; the per-shard brute force is not it)
    if nrec > 0:
        tb0 = time.perf_counter()
        truth = g.bruteforce_topk(dq, nrec, K)
        brute_s = time.perf_counter() - tb0
        brute_ms = g.last_launch()["last_kernel_ms"]
        recall = recall_of(out_ids[:nrec], truth, K)

    sweep = []
    if args.ef_sweep == "auto":
        args.ef_sweep = "128,256,512" if world == 1 and not streamed else ""
    for ef2 in [int(x) for x in args.ef_sweep.split(",") if x]:
        run_steps(1, ef2)
        g.sync()
        k2, nd2, ne2 = run_steps(3, ef2, collect=True)
        o2 = np.empty((NQ, K), np.int64)
        g.dev_download(o2, d_ids)
        r2 = recall_of(o2[:nrec], truth, K) if nrec else None
        ab2 = nd2 * D * 4 + ne2 * (2 * M) * 4 + nd2 * 4
        sweep.append({"ef": ef2, "queries_per_s": NQ / (np.mean(k2) * 1e-3), "recall_at_k": r2, "n_dist_per_query": nd2 / NQ,
                      "kernel_ms": float(np.mean(k2)), "roofline_frac": ab2 / (np.mean(k2) * 1e-3) / 1e9 / HBM_PEAK_GBS})

    # ---- one query per call (the SQL surface's shape: mn_hnsw_search from host memory, answer back in host memory): the
    #      latency-bound kernel (k_beam_coop, queues in registers), not the throughput one ----
    lone = None
    if rank == 0 and world == 1 and not sharded and not streamed:
        nl = min(200, NQ)
        for i in range(20):
            g.search(Q[i], K, EF)
        tl, same_ids, same_bits = [], 0, 0
        out_ds = np.empty((NQ, K), np.float32)
        g.dev_download(out_ds, d_ds)
        if sweep:  # the ef sweep left its last ef's answers in the buffers: bring the headline ef's back
            run_steps(1, EF)
            g.sync()
            g.dev_download(out_ids, d_ids)
            g.dev_download(out_ds, d_ds)
        for i in range(nl):
            t1 = time.perf_counter()
            li, ld_ = g.search(Q[i], K, EF)
            tl.append((time.perf_counter() - t1) * 1e3)
            same_ids += int(np.array_equal(li, out_ids[i][:len(li)]))
            same_bits += int(np.array_equal(np.asarray(ld_, np.float32).view(np.int32), out_ds[i][:len(li)].view(np.int32)))
        lone = {"queries": nl, "ms_per_query_median": float(np.median(tl)), "ms_per_query_p90": float(np.percentile(tl, 90)),
                "same_ids_as_the_batch_kernel": same_ids, "same_distance_bits_as_the_batch_kernel": same_bits,
                "what": "mn_hnsw_search, one query per call, host memory in and out (k_beam_coop; the batch above is k_beam)"}

    # ---- CPU baseline: the oracle (single-threaded port of the reference algorithm) on the SAME
    #      graph and the SAME queries; also a full-size parity check of the returned ids ----
    cpu = None
    parity = None
    build_cpu = None
    exact_at_size = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_queries > 0:  # N=1 only (contract)
        from oracle import orc

        progress("cpu baseline: oracle, then the compiled reference, on the same graph")
        o = orc.Oracle(D, args.metric, M, EFC, order=orc.ORDER_SSE if args.order == "sse" else orc.ORDER_WAVE)
        o.load_from_device(g, vectors=X)
        nc = min(args.cpu_queries, NQ)
        tc = time.perf_counter()
        oi, od, oc = o.search_many(Q[:nc], K, EF)
        cpu_s = time.perf_counter() - tc
        run_steps(1, EF, merge=False)  # rank 0 only: no collective here
        g.sync()
        g.dev_download(out_ids, d_ids)
        gd = np.empty((NQ, K), np.float32)
        g.dev_download(gd, d_ds)
        parity = {"queries": nc, "ids_identical": bool(np.array_equal(oi, out_ids[:nc])),
                  "dists_bit_identical": bool(np.array_equal(od.view(np.int32), gd[:nc].view(np.int32)))}
        if args.order == "wave":
            # the fast order is checked bit-for-bit against its own CPU restatement above; here the same graph is
            # searched in the REFERENCE's summation order and id-set mismatches are counted (north_star: bit-exact
            # id sets, distances within 1e-5 relative)
            nr = min(nc, 500)
            o2 = orc.Oracle(D, args.metric, M, EFC, order=orc.ORDER_SSE)
            o2.load_from_device(g, vectors=X)
            ri, rd, rc = o2.search_many(Q[:nr], K, EF)
            same = [set(ri[i].tolist()) == set(out_ids[i].tolist()) for i in range(nr)]
            rel = np.abs(rd - gd[:nr]) / np.maximum(np.abs(rd), 1.0)
            parity["vs_reference_order"] = {"queries": nr, "id_sets_identical": int(sum(same)),
                                            "max_rel_distance_diff": float(rel.max())}
        cpu = {"value": nc / cpu_s, "unit": "queries/s", "cores": 1, "kind": "port",
               "sample": f"{nc} of the {NQ} queries (same graph, k={K}, ef={EF}), oracle/mn_oracle.c single thread, "
                         f"bitmap visited set; {cpu_s:.1f}s of CPU work"}
        if orc.have_ref() and args.ref_queries > 0:
            # the reference's own compiled hnsw_search (oracle/_ref, built from /root/reference/src in the build
            # container) on the same graph, loaded through the reference's own shadow-table load API
            r = orc.Ref(D, args.metric, M, EFC)
            r.load_from_device(g, vectors=X)
            nr = min(args.ref_queries, nc)
            tr = time.perf_counter()
            ri, rd, rc = r.search_many(Q[:nr], K, EF)
            ref_s = time.perf_counter() - tr
            parity["vs_reference_binary"] = {
                "queries": nr,
                "id_sets_identical": int(sum(set(ri[i].tolist()) == set(out_ids[i].tolist()) for i in range(nr))),
                "ids_identical": bool(np.array_equal(ri, out_ids[:nr])),
                "dists_bit_identical": bool(np.array_equal(rd.view(np.int32), gd[:nr].view(np.int32))),
                "max_rel_distance_diff": float((np.abs(rd - gd[:nr]) / np.maximum(np.abs(rd), 1.0)).max())}
            cpu = {"value": nr / ref_s, "unit": "queries/s", "cores": 1, "kind": "reference",
                   "sample": f"{nr} of the {NQ} queries (same graph, k={K}, ef={EF}) through the reference's own "
                             f"hnsw_search (src/hnsw_algo.c, gcc -O2, linear visited set), single thread; "
                             f"{ref_s:.1f}s of CPU work",
                   "port": {"value": nc / cpu_s, "unit": "queries/s", "cores": 1,
                            "sample": f"{nc} queries, oracle/mn_oracle.c (bitmap visited set), {cpu_s:.1f}s"}}
            if args.exact_inserts > 0:
                # the build half: the SAME new vectors inserted one at a time (reference semantics) into the SAME full-size
                # graph by the compiled reference (1 core) and by the device (MN_BUILD_SEQUENTIAL); both level streams are
                # re-seeded alike, so the two graphs must stay identical: every new node's lists are compared
                ni = args.exact_inserts
                Xn = gen_vectors(ni, D, 4444, args.dataset)
                idn = np.arange(N + 1, N + 1 + ni, dtype=np.int64)
                r.R.hnsw_seed_rng(r.h, 987654321)
                g.seed_rng(987654321)
                tr = time.perf_counter()
                rc_ref = r.insert_many(idn, Xn)
                ref_ins_s = time.perf_counter() - tr
                tg = time.perf_counter()
                rc_gpu = g.insert_batch(idn, Xn, pkg.BUILD_SEQUENTIAL)
                g.sync()
                gpu_ins_s = time.perf_counter() - tg
                same = rc_ref == 0 and rc_gpu == 0 and r.graph(idn) == g.graph(idn)
                build_cpu = {"value": ni / ref_ins_s, "unit": "vectors/s", "cores": 1, "kind": "reference",
                             "sample": f"{ni} hnsw_insert calls of the compiled reference (src/hnsw_algo.c, gcc -O2) into the "
                                       f"{N}-node graph built on the GPU, loaded through the reference's own load API; "
                                       f"{ref_ins_s:.1f}s of CPU work"}
                exact_at_size = {"inserts": ni, "gpu_exact_vectors_per_s": ni / gpu_ins_s,
                                 "reference_vectors_per_s": ni / ref_ins_s,
                                 "new_nodes_lists_identical_to_reference": bool(same)}
            del r

    # HBM traffic per launch from the committed PMC passes (profiles/traffic.json), when this exact workload was profiled AND
    # the kernel's sources are still the ones it was measured on (the entry carries their sha-256)
    traffic, traffic_note = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        key = f"{N}x{D}_{args.dataset}_{args.order}_nq{NQ}_k{K}_ef{EF}"
        if key in tj:
            ent = tj[key]
            now = kernel_sources_sha(ent.get("kernel_sources", []))
            if ent.get("kernel_sources_sha256") and now == ent["kernel_sources_sha256"]:
                traffic = ent["traffic_bytes"]
                traffic_note = f"rocprofv3 --pmc passes of round {ent.get('measured_in_round')} at commit {ent.get('measured_at_commit')}; " \
                               f"kernel sources unchanged since (sha-256 {now[:12]})"
            else:
                traffic_note = "the committed PMC measurement is for other kernel sources than the ones built here: not reported"
    except (OSError, ValueError):
        pass

    at_target = None
    quality = None
    wave_leg = None
    if rank == 0 and world == 1 and not streamed and args.order == "sse" and not args.no_wave_leg and not args.no_cpu_baseline:
        progress("wave-order leg: same data, MN_ORDER_WAVE, id sets vs the compiled reference")
        g.close()
        g = None
        wave_leg = wave_order_leg(pkg, args, dev_ord, M, EFC, X, Q)
    if rank == 0 and world == 1 and args.recall_target > 0 and args.dataset == "gaussian":
        if g is not None:
            g.close()  # make room: the second index is the same size
        g = None
        X = None
        at_target = recall_target_leg(pkg, args, dev_ord, order, M, EFC, args.recall_target,
                                      [d for d in args.target_datasets.split(",") if d])
    if rank == 0 and world == 1 and args.quality_n > 0:
        quality = graph_quality_leg(pkg, args, dev_ord, order, M, EFC, min(args.quality_n, N),
                                    [d for d in args.quality_datasets.split(",") if d], (128, 256))

    graph = None
    if rank == 0 and world == 1 and not streamed and not args.no_graph_block and not args.no_cpu_baseline:
        if g is not None:
            g.close()
        g = None
        X = None
        graph = graph_block(pkg, args, dev_ord, t_run0)

    if rank == 0:
        total_q = NQ * args.steps * (1 if sharded else world)
        line = {
            "metric": "kNN queries/sec (10k-query batch, k=10, ef=128, 1M x 768 f32 HNSW index) + recall@10",
            "value": total_q / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{N}x{D} f32 {args.dataset}, HNSW M={M} efC={EFC} {args.metric}, build on GPU + "
                                   f"{NQ}-query batched kNN k={K} ef={EF}",
                       "n": N, "dim": D, "nq": NQ, "k": K, "ef": EF, "order": args.order, "dataset": args.dataset,
                       "parallelism": (f"sharded index (rowid mod N) + all-gather of per-shard top-k ({args.backend}) + device merge, below the C-ABI" if sharded else
                                       "replica per GPU, queries sharded") if world > 1 else "single GPU"},
            "recall_at_10": recall,
            "recall_queries": nrec,
            "n_dist_per_query": n_dist / NQ,
            "build_vectors_per_s": N / build_s,
            # sharded index: every rank builds its own N-vector shard (config 3) → aggregate; replicas: the same graph is
            # built once per GPU, which does not scale (a single sequential-semantics graph does not shard)
            "build_vectors_per_s_all_gpus": (N * world if sharded else N) / build_s_max,  # jointly built graph: N / time
            "build_s": build_s,
            "build_exact_at_full_size": exact_at_size,
            # build side of the roofline: the search half (k_beam<BUILD>) is the dominant kernel of a build; algorithmic bytes
            # as for search (candidate rows + link rows + visited probes), counted on the device
            "build_roofline": (lambda ab: {"bound": "hbm", "kernel": "k_beam<BUILD>", "kernel_ms_total": bst["search_ms"],
                                           "link_ms_total": bst["link_ms"], "batches": bst["batches"],
                                           "n_dist_per_insert": bst["n_dist"] / max(1, bst["nodes"]),
                                           "algorithmic_bytes": ab, "achieved": ab / max(bst["search_ms"], 1e-9) / 1e6,
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": ab / max(bst["search_ms"], 1e-9) / 1e6 / HBM_PEAK_GBS,
                                           "search_share_of_build_wall": bst["search_ms"] * 1e-3 / build_s})(
                bst["n_dist"] * D * 4 + bst["n_expanded"] * (2 * M) * 4 + bst["n_dist"] * 4) if bst["batches"] else None,
            "build_cpu_baseline": build_cpu,
            "graph_quality_exact_vs_batched": quality,
            "wave_order": wave_leg,
            "build_mode": "batch-synchronous (batch <= max(1, n/16), cap 8192); " +
                          ("one graph built jointly: search half of each batch split over the GPUs, selected lists all-gathered, "
                           "every replica links (bit-identical to the 1-GPU build)" if shared_build else
                           "one shard per GPU" if sharded else "single GPU"),
            "parity_vs_oracle": parity,
            "ef_sweep": sweep,
            "one_query_per_call": lone,
            "at_recall_target": at_target,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note, "kernel": "k_beam",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         # also against what a float4 copy kernel reaches on this part
                         "frac_of_measured_copy_bw": achieved / HBM_COPY_GBS, "measured_copy_bw": HBM_COPY_GBS},
            "ground_truth": None if recall is None else {
                "kernel": "k_brute_mfma (v_mfma_f32_32x32x2_f32 GEMM tile + fused top-k)", "queries": nrec,
                "kernel_ms": brute_ms, "wall_s": brute_s,
                "tflops": 2.0 * nrec * N * D / (brute_ms * 1e-3) / 1e12 if brute_ms else None},
            "cpu_baseline": dict(cpu, host=host_cpu()) if cpu else None,
            # configs 4 and 5 (Node2Vec, Leiden): bench_graph.py's lines for them, measured in this same run
            "graph": graph,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()
    if g is not None:
        g.close()


if __name__ == "__main__":
    main()
