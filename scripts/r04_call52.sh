#!/bin/bash
# round-4 GPU call 52: PageRank range launches with balanced gathers (range-major pieces, shares parked in LDS: k_pr_pull_flat)
# beside the one-lane-per-target range kernel, one box, interleaved; parity tests; f-4 counter passes on this mn_graph_algo.hip
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 300 python -m pytest tests/test_graph_tvf.py -m gpu -x -q > $O/t_call52.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/t_call52.log
[ $rc -eq 0 ] || exit $rc
for F in 1 0 1 0; do
  MN_PR_FLAT=$F timeout -k 10 100 python bench_graph.py --workload pagerank --no-ref-sql > $O/pr_f$F.json 2> $O/pr_f$F.err; echo -n "flat=$F rc=$? "
  python -c "
import json
d=json.loads(open('$O/pr_f$F.json').read().strip().splitlines()[-1])
print(round(d['config']['device_ms'],2), round(d['roofline']['frac'],4), round(d['config']['wall_ms'],1))"
done | tee $O/ab_pagerank_flat.txt
rm -rf $O/prof_tvf_*
bash scripts/prof_tvf.sh
cd "$R"
KS=$(ls $O/prof_tvf_k/*kernel_stats.csv 2>/dev/null | head -1)
FC=$(ls $O/prof_tvf_f/*counter_collection.csv 2>/dev/null | head -1)
WC=$(ls $O/prof_tvf_w/*counter_collection.csv 2>/dev/null | head -1)
cp "$FC" $O/tvf_fetch_counters.csv; cp "$WC" $O/tvf_write_counters.csv
python scripts/summarize_prof.py r04_tvf_1M_20M "$KS" "$FC" "$WC" && cp profiles/r04_tvf_1M_20M_* $O/ && grep "k_pr_" profiles/r04_tvf_1M_20M_pmc_summary.csv
