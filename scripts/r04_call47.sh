#!/bin/bash
# round-4 GPU call 47: Brandes workgroup width on the cell layout (4 / 8 / 16 / 32 lanes), one box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 300 python -m pytest tests/test_graph_tvf.py -m gpu -x -q -k betweenness > $O/t_call47.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/t_call47.log
[ $rc -eq 0 ] || exit $rc
for L in 4 16 8 32 4 16; do
  MN_BRANDES_LANES=$L timeout -k 10 100 python bench_graph.py --workload betweenness --no-ref-sql > $O/bc_l$L.json 2> $O/bc_l$L.err; echo -n "lanes=$L rc=$? "
  python -c "
import json
d=json.loads(open('$O/bc_l$L.json').read().strip().splitlines()[-1])
print(round(d['config']['device_ms'],1), round(d['at_published_size_through_sql']['this_extension_ms'],1))"
done | tee $O/ab_brandes_lanes.txt
