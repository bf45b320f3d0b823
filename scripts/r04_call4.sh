#!/bin/bash
# round-4 GPU call 4: quad loads A/B incl. build time, node2vec (pipelined walk_grad, index leg with the incremental link step),
# then the whole GPU suite on the new kernels
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
bash scripts/ab_search2.sh sse quad0.so quad12.so > $O/ab_quad2.log 2>&1; cat $O/ab_quad2.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_n2v2_k -o n2v -- python3 $R/bench_graph.py --workload node2vec --steps 1 --warmup 0 > $O/n2v_bench3.json 2> $O/n2v_bench3.err; echo "n2v rc=$?"
cd "$R"
python - <<'PY'
import json
d=json.load(open("gpurun_out/n2v_bench3.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], json.dumps(d["parity_vs_oracle"]))
print(json.dumps(d["to_hnsw_index"]))
PY
head -8 $O/prof_r04_n2v2_k/n2v_kernel_stats.csv | cut -c1-150
python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; echo "all rc=$?"; tail -5 $O/t_all.log
