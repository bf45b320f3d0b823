#!/bin/bash
# round-4 GPU call 8: link step with settled tie state (counters on node2vec embeddings, hub parity tests), node2vec bench on the
# ER graph (index leg) and on the Barabasi-Albert graph
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
MN_AB_LIB=build/ab/linkdbg.so python scripts/probe_link.py 300000 -1 > $O/link_dbg3.log 2>&1; cat $O/link_dbg3.log
python -m pytest tests/test_gpu_hnsw.py tests/test_schedule_pins.py -m gpu -x -q > $O/t_call8.log 2>&1; echo "hnsw rc=$?"; tail -3 $O/t_call8.log
python bench_graph.py --workload node2vec --steps 1 --warmup 0 > $O/n2v_bench4.json 2> $O/n2v_bench4.err; echo "n2v rc=$?"
python bench_graph.py --workload node2vec --steps 1 --warmup 0 --n2v-model ba --no-index-leg > $O/n2v_bench_ba.json 2> $O/n2v_bench_ba.err; echo "n2v ba rc=$?"
python - <<'PY'
import json
for f in ("n2v_bench4", "n2v_bench_ba"):
    d=json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["degree"], json.dumps(d["to_hnsw_index"])[:700])
PY
