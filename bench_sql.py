#!/usr/bin/env python3
"""The drop-in user's view (BASELINE.json configs[0]): N x dim vectors INSERTed one row at a time through the
`hnsw_index` virtual table, then one kNN query per SELECT — the published methodology of the reference's own harness
(benchmarks/harness/treatments/vss.py:262-319: per-row `INSERT INTO t(rowid, vector)`, per-query
`SELECT rowid, distance ... WHERE vector MATCH ? AND k = ? AND ef_search = ?`, one transaction, recall vs brute force).

Prints ONE JSON line: insert rate (vectors/s) and search latency (ms/query) for each MUNINN_HNSW_MODE of this
extension, with the reference's own extension (oracle/_ref/muninn.so, compiled from the reference's sources in the build
container; it travels to the GPU box as a binary) timed on the same inputs on the host CPU beside it.

    python bench_sql.py [--n 10000 --dim 128]
"""
import argparse
import json
import os
import sqlite3
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
EXT = os.path.join(ROOT, "sqlite-muninn_amd", "ext", "muninn")
REF = os.path.join(ROOT, "oracle", "_ref", "muninn")


def run(ext, X, Q, truth, m, efc, k, ef, metric):
    c = sqlite3.connect(":memory:")
    c.enable_load_extension(True)
    c.load_extension(ext)
    c.execute(f"CREATE VIRTUAL TABLE bench_vec USING hnsw_index(dimensions={X.shape[1]}, metric='{metric}', m={m}, ef_construction={efc})")
    t0 = time.perf_counter()
    for i in range(len(X)):
        c.execute("INSERT INTO bench_vec (rowid, vector) VALUES (?, ?)", (i + 1, X[i].tobytes()))
    c.execute("SELECT rowid FROM bench_vec WHERE rowid = 1").fetchall()  # deferred / fast modes flush at the first read
    ins = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = []
    for q in Q:
        res.append({r[0] for r in c.execute("SELECT rowid, distance FROM bench_vec WHERE vector MATCH ? AND k = ? AND ef_search = ?",
                                            (q.tobytes(), k, ef))})
    sea = time.perf_counter() - t0
    c.commit()
    c.close()
    recall = float(np.mean([len(res[i] & truth[i]) / k for i in range(len(Q))]))
    return {"insert_rate_vps": len(X) / ins, "search_latency_ms": sea / len(Q) * 1e3, "recall_at_k": recall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--ef", type=int, default=64)
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--ref-n", type=int, default=3000, help="rows the reference extension is timed on (its insert is ~100 rows/s)")
    args = ap.parse_args()
    subprocess.run(["make", "-s", "-C", os.path.dirname(EXT)], check=True)
    rng = np.random.default_rng(42)
    X = rng.standard_normal((args.n, args.dim), dtype=np.float32)
    Q = rng.standard_normal((args.queries, args.dim), dtype=np.float32)
    k, m, efc = 10, 16, 200

    def truth_of(Xs):
        D = ((Q[:, None, :] - Xs[None, :, :]) ** 2).sum(2) if args.metric == "l2" else -(Q @ Xs.T)
        return [set((np.argsort(D[i], kind="stable")[:k] + 1).tolist()) for i in range(len(Q))]

    truth = truth_of(X)
    out = {"metric": "hnsw_index through SQL: per-row INSERT rate and per-query SELECT latency (the reference harness's methodology)",
           "unit": "vectors/s, ms/query", "data": "synthetic",
           "config": {"workload": f"{args.n}x{args.dim} f32 gaussian, {args.metric}, m={m} ef_construction={efc} k={k} ef_search={args.ef}, "
                                  f"{args.queries} queries, one connection, rows inserted one statement at a time"},
           "modes": {}}
    for mode in ("exact", "deferred", "fast"):
        os.environ["MUNINN_HNSW_MODE"] = mode
        out["modes"][mode] = run(EXT, X, Q, truth, m, efc, k, args.ef, args.metric)
    os.environ.pop("MUNINN_HNSW_MODE")
    if os.path.exists(REF + ".so"):
        nr = min(args.ref_n, args.n)
        r = run(REF, X[:nr], Q, truth_of(X[:nr]), m, efc, k, args.ef, args.metric)
        r["sample"] = f"the reference's own extension (oracle/_ref/muninn.so, gcc -O2) on the first {nr} rows, host CPU, 1 core"
        out["cpu_baseline"] = dict(r, kind="reference", cores=1)
        os.environ["MUNINN_HNSW_MODE"] = "exact"
        out["same_rows_exact_mode"] = run(EXT, X[:nr], Q, truth_of(X[:nr]), m, efc, k, args.ef, args.metric)
        os.environ.pop("MUNINN_HNSW_MODE")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    sys.exit(main())
