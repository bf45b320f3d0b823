#!/bin/bash
# round-4 GPU call 50: PageRank range launches with the first W entries of a piece in one predicated batch (W = 0 / 8 / 12 / 16,
# one box, interleaved), parity tests, then the f-4 counter passes on the final mn_graph_algo.hip
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 300 python -m pytest tests/test_graph_tvf.py -m gpu -x -q > $O/t_call50.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/t_call50.log
[ $rc -eq 0 ] || exit $rc
for W in 0 12 8 16 0 12 8 16; do
  MN_PR_FIRST=$W timeout -k 10 100 python bench_graph.py --workload pagerank --no-ref-sql > $O/pr_w$W.json 2> $O/pr_w$W.err; echo -n "first=$W rc=$? "
  python -c "
import json
d=json.loads(open('$O/pr_w$W.json').read().strip().splitlines()[-1])
print(round(d['config']['device_ms'],2), round(d['roofline']['frac'],4))"
done | tee $O/ab_pagerank_first.txt
rm -rf $O/prof_tvf_*
bash scripts/prof_tvf.sh
cd "$R"
KS=$(ls $O/prof_tvf_k/*kernel_stats.csv 2>/dev/null | head -1)
FC=$(ls $O/prof_tvf_f/*counter_collection.csv 2>/dev/null | head -1)
WC=$(ls $O/prof_tvf_w/*counter_collection.csv 2>/dev/null | head -1)
cp "$FC" $O/tvf_fetch_counters.csv; cp "$WC" $O/tvf_write_counters.csv
python scripts/summarize_prof.py r04_tvf_1M_20M "$KS" "$FC" "$WC" && cp profiles/r04_tvf_1M_20M_* $O/ && grep "k_pr_" profiles/r04_tvf_1M_20M_pmc_summary.csv
