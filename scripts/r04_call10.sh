#!/bin/bash
# round-4 GPU call 10 (evidence, part 2): PMC + kernel-trace passes of the headline bench on the final kernels, the 128-d bench
# line, the SQL surface like for like (the reference extension on ALL rows), the f-4 table-valued functions
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
bash scripts/prof_bench.sh r04_bench_1Mx768_sse | tail -4
cd "$R"
python bench.py --dim 128 --no-wave-leg --recall-target 0 --quality-n 0 --no-graph-block --exact-inserts 200 --steps 10 > $O/bench_128d.json 2> $O/bench_128d.err; echo "bench128 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_128d.json"))
print("128-d:", d["value"], d["roofline"]["frac"], d["build_vectors_per_s"], d["build_roofline"]["frac"])
PY
python bench_sql.py --n 10000 --dim 128 --ref-n 10000 > $O/sql_10kx128.json 2> $O/sql_10kx128.err; echo "sql1 rc=$?"
python bench_sql.py --n 10000 --dim 768 --ref-n 10000 > $O/sql_10kx768.json 2> $O/sql_10kx768.err; echo "sql2 rc=$?"
python bench_sql.py --n 3000 --dim 128 --ref-n 3000 > $O/sql_3kx128.json 2> $O/sql_3kx128.err; echo "sql3 rc=$?"
python bench_graph.py --workload tvf > $O/tvf_bench.json 2> $O/tvf_bench.err; echo "tvf rc=$?"; cut -c1-600 $O/tvf_bench.json
