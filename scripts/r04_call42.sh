#!/bin/bash
# round-4 GPU call 42: Brandes with the per-node state in one 32-byte cell (parity tests, bench leg with the reference's rows beside it);
# PageRank by source ranges: range size (2^18 .. 2^15 nodes) x non-temporal loads of the streamed operands, on one box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 500 python -m pytest tests/test_graph_tvf.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/t_call42.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/t_call42.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench_graph.py --workload betweenness > $O/tvf_betweenness_cells.json 2> $O/tvf_betweenness_cells.err; echo "bc rc=$?"
python -c "
import json
d=json.loads(open('$O/tvf_betweenness_cells.json').read().strip().splitlines()[-1])
print('cells', d['config']['device_ms'], d['at_published_size_through_sql']['this_extension_ms'], d['at_published_size_through_sql']['rows_equal_to_reference'])"
for L in 18 17 16 15; do for NT in 1 0; do
  MN_PR_TILE_LOG2=$L MN_PR_NT=$NT timeout -k 10 200 python bench_graph.py --workload pagerank --no-ref-sql > $O/pr_${L}_${NT}.json 2> $O/pr_${L}_${NT}.err; echo -n "pr log2=$L nt=$NT rc=$? "
  python -c "
import json
d=json.loads(open('$O/pr_${L}_${NT}.json').read().strip().splitlines()[-1])
print(d['config']['device_ms'], d['roofline']['frac'])"
done; done | tee $O/ab_pagerank_ranges.txt
