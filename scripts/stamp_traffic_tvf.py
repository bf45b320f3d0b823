#!/usr/bin/env python3
"""profiles/traffic.json entries for the three f-4 workloads of bench_graph.py (--workload tvf --no-ref-sql) from the raw
rocprofv3 counter files of scripts/prof_tvf.sh: FETCH_SIZE (doubled: gfx950, MI355X_MICROARCH.md) + WRITE_SIZE, summed over the
launches of ONE timed run at the C-ABI size (the warm-up run and the 10 000-node SQL run are told apart by grid size and order).
usage: stamp_traffic_tvf.py <fetch counter csv> <write counter csv> <commit> [<summary tag> [key substring ...]]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402


def rows(path, cname):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        k = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        out.append((int(r["Start_Timestamp"]), k, int(r["Grid_Size"]), float(r["Counter_Value"])))
    out.sort()
    return out


def one_run(rs, names, runs_at_big_size, iters_of_timed_run=None):
    """KB of one timed run: launches of `names` at the largest grid seen for the first of them"""
    mine = [r for r in rs if r[1] in names]
    if not mine:
        return None, 0
    lead = next(nme for nme in names if any(r[1] == nme for r in mine))
    big = max(r[2] for r in mine if r[1] == lead)
    # a launch belongs to the big case when the lead kernel launched last before or at it had the big grid
    total, n, cur_big = 0.0, 0, False
    picked = []
    for r in mine:
        if r[1] == lead:
            cur_big = r[2] == big
        if cur_big:
            picked.append(r)
    if iters_of_timed_run is not None:  # pagerank: 2 warm-up iterations, then the timed 100 — keep the last 100/102 of each kernel
        keep = []
        for nme in names:
            mine_k = [r for r in picked if r[1] == nme]
            if mine_k:
                keep += mine_k[-max(1, round(len(mine_k) * iters_of_timed_run / (iters_of_timed_run + 2))):]
        picked, runs_at_big_size = keep, 1
    for r in picked:
        total += r[3]
        n += 1
    return total / runs_at_big_size, n // runs_at_big_size


def main():
    fpath, wpath, commit = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r04_tvf_1M_20M"
    only = sys.argv[5:]
    f, w = rows(fpath, "FETCH_SIZE"), rows(wpath, "WRITE_SIZE")
    tj_path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tj_path))
    spec = {
        "tvf_pagerank_er_1000000_nodes_avg_degree_20": (["k_pr_pull_flat", "k_pr_pull_tile", "k_pr_pull", "k_pr_share"], 1, 100, ["mn_graph_algo.hip"]),
        "tvf_components_er_1000000_nodes_avg_degree_20": (["k_cc_hook", "k_cc_root", "k_cc_out", "k_uf_init"], 2, None, ["mn_graph_algo.hip"]),
        "tvf_betweenness_er_20000_nodes_avg_degree_20": (["k_brandes_accumulate", "k_brandes_sources"], 2, None, ["mn_graph.hip"]),
    }
    for key, (names, runs, iters, srcs) in spec.items():
        if only and not any(o in key for o in only):
            continue
        if key.startswith("tvf_betweenness"):
            # sources launches precede their accumulate launch: classify by the accumulate grid that FOLLOWS → walk backwards
            f2, w2 = [(-t, k, g, v) for t, k, g, v in f], [(-t, k, g, v) for t, k, g, v in w]
            f2.sort()
            w2.sort()
            fk, nf = one_run(f2, names, runs)
            wk, _ = one_run(w2, names, runs)
        else:
            fk, nf = one_run(f, names, runs, iters)
            wk, _ = one_run(w, names, runs, iters)
        if fk is None or wk is None:
            print(key, "not in the counter files")
            continue
        tj[key] = {"kernel": " + ".join(names) + f", the {nf} launches of one timed run (bench_graph.py --workload tvf --no-ref-sql)",
                   "fetch_size_kb": fk, "write_size_kb": wk, "traffic_bytes": int((2 * fk + wk) * 1024),
                   "source": f"profiles/{tag}_pmc_summary.csv (round 4, scripts/prof_tvf.sh: separate --pmc passes; FETCH_SIZE doubled); "
                             "per-run sums taken from the raw counter files by scripts/stamp_traffic_tvf.py",
                   "kernel_sources": srcs, "kernel_sources_sha256": kernel_sources_sha(srcs), "measured_in_round": 4,
                   "measured_at_commit": commit}
        print(key, tj[key]["traffic_bytes"], nf)
    json.dump(tj, open(tj_path, "w"), indent=1)


if __name__ == "__main__":
    main()
