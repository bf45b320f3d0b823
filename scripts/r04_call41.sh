#!/bin/bash
# round-4 GPU call 41: PageRank pulled one source range per launch (k_pr_pull_tile) — parity tests, A/B against the one-launch
# kernel on the same box, kernel stats + FETCH/WRITE passes of the pagerank workload
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 400 python -m pytest tests/test_graph_tvf.py -m gpu -x -q > $O/t_call41.log 2>&1; rc=$?; echo "tvf tests rc=$rc"; tail -3 $O/t_call41.log
[ $rc -eq 0 ] || exit $rc
for V in "tiles:" "one_launch:MN_PR_TILES=0" "tiles_again:" "one_launch_again:MN_PR_TILES=0"; do
  T=${V%%:*}; E=${V#*:}
  env $E timeout -k 10 200 python bench_graph.py --workload pagerank --no-ref-sql > $O/pr_$T.json 2> $O/pr_$T.err; echo "pr $T rc=$?"
  python -c "
import json
d=json.loads(open('$O/pr_$T.json').read().strip().splitlines()[-1])
print('$T', d['config']['device_ms'], d['at_published_size_through_sql']['this_extension_ms'], d['roofline']['frac'])"
done
cd /tmp && export TMPDIR=/tmp
for P in k f w; do
  case $P in k) OPT="--kernel-trace --stats";; f) OPT="--pmc FETCH_SIZE";; w) OPT="--pmc WRITE_SIZE";; esac
  timeout -k 5 200 rocprofv3 $OPT --output-format csv -d $O/prof_pr_$P -o pr -- python3 $R/bench_graph.py --workload pagerank --no-ref-sql > $O/prof_pr_$P.log 2>&1; echo "$P rc=$?"
done
cd "$R"
KS=$(ls $O/prof_pr_k/*kernel_stats.csv $O/prof_pr_k/*/*kernel_stats.csv 2>/dev/null | head -1)
FC=$(ls $O/prof_pr_f/*counter_collection.csv $O/prof_pr_f/*/*counter_collection.csv 2>/dev/null | head -1)
WC=$(ls $O/prof_pr_w/*counter_collection.csv $O/prof_pr_w/*/*counter_collection.csv 2>/dev/null | head -1)
cp "$FC" $O/pr_fetch_counters.csv; cp "$WC" $O/pr_write_counters.csv
python scripts/summarize_prof.py r04_pagerank_tiles_1M_20M "$KS" "$FC" "$WC" && cp profiles/r04_pagerank_tiles_1M_20M_* $O/ && cat profiles/r04_pagerank_tiles_1M_20M_pmc_summary.csv
