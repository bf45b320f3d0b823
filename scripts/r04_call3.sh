#!/bin/bash
# round-4 GPU call 3: link step for hub targets (parity + the node2vec -> index leg), Leiden occupancy, SSE-order quad loads A/B
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
python -m pytest tests/test_leiden.py tests/test_gpu_hnsw.py tests/test_schedule_pins.py tests/test_node2vec.py tests/test_parallel.py -m gpu -x -q > $O/t_call3.log 2>&1; echo "pytest rc=$?"; tail -4 $O/t_call3.log
python scripts/probe_leiden.py 3 > $O/lei_u.log 2>&1; cat $O/lei_u.log
MN_LEIDEN_FULL_TABLES=1 python scripts/probe_leiden.py 3 > $O/lei_u_full.log 2>&1; cat $O/lei_u_full.log
bash scripts/ab_search2.sh sse quad4.so quad8.so > $O/ab_quad.log 2>&1; cat $O/ab_quad.log
python bench_graph.py --workload node2vec --steps 1 --warmup 0 > $O/n2v_bench2.json 2> $O/n2v_bench2.err; echo "n2v rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/n2v_bench2.json"))
print(json.dumps(d["to_hnsw_index"]))
PY
