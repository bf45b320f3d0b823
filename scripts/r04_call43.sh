#!/bin/bash
# round-4 GPU call 43: k_cc_root with one atomic per distinct root of a wavefront; PageRank source ranges of 2^19 beside 2^18;
# the whole f-4 bench (with the compiled reference's TVFs beside it) and its kernel stats + FETCH/WRITE passes on the final kernels
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 400 python -m pytest tests/test_graph_tvf.py -m gpu -x -q > $O/t_call43.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/t_call43.log
[ $rc -eq 0 ] || exit $rc
for L in 18 19 18 19; do
  MN_PR_TILE_LOG2=$L timeout -k 10 200 python bench_graph.py --workload pagerank --no-ref-sql > $O/pr_${L}.json 2> $O/pr_${L}.err; echo -n "pr log2=$L rc=$? "
  python -c "
import json
d=json.loads(open('$O/pr_${L}.json').read().strip().splitlines()[-1])
print(d['config']['device_ms'], d['roofline']['frac'])"
done | tee $O/ab_pagerank_19.txt
timeout -k 10 400 python bench_graph.py --workload tvf > $O/tvf_bench.json 2> $O/tvf_bench.err; echo "tvf rc=$?"; cut -c1-420 $O/tvf_bench.json
bash scripts/prof_tvf.sh
cd "$R"
KS=$(ls $O/prof_tvf_k/*kernel_stats.csv 2>/dev/null | head -1)
FC=$(ls $O/prof_tvf_f/*counter_collection.csv 2>/dev/null | head -1)
WC=$(ls $O/prof_tvf_w/*counter_collection.csv 2>/dev/null | head -1)
cp "$FC" $O/tvf_fetch_counters.csv; cp "$WC" $O/tvf_write_counters.csv
python scripts/summarize_prof.py r04_tvf_1M_20M "$KS" "$FC" "$WC" && cp profiles/r04_tvf_1M_20M_* $O/ && cat profiles/r04_tvf_1M_20M_pmc_summary.csv
