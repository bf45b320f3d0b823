#!/bin/bash
# round-4 GPU call 9 (evidence, part 1): f-4 parity on the new kernels, PMC + kernel-trace passes of the headline bench and of
# Leiden (unweighted / weighted), the node2vec pipeline's kernel trace, the 128-d bench line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
python -m pytest tests/test_graph_tvf.py -m gpu -x -q > $O/t_call9.log 2>&1; echo "tvf rc=$?"; tail -3 $O/t_call9.log
bash scripts/prof_bench.sh r04_bench_1Mx768_sse | tail -4
bash scripts/prof_leiden.sh "" r04u | tail -3
bash scripts/prof_leiden.sh weighted r04w | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_n2v_final_k -o n2v -- python3 $R/bench_graph.py --workload node2vec --steps 1 --warmup 0 > $O/n2v_bench_final.json 2> $O/n2v_bench_final.err; echo "n2v rc=$?"
cd "$R"
python bench.py --dim 128 --no-wave-leg --recall-target 0 --quality-n 0 --no-graph-block --exact-inserts 200 --steps 10 > $O/bench_128c.json 2> $O/bench_128c.err; echo "bench128 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_128c.json"))
print("128-d:", d["value"], d["roofline"]["frac"], d["build_vectors_per_s"], d["build_roofline"]["frac"])
PY
