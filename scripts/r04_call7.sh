#!/bin/bash
# round-4 GPU call 7: link-step counters on real node2vec embeddings; node2vec all-to-all exchange (2- and 4-rank parity);
# quad-12 loads with the short-row chain: HNSW parity at every dimension, 128-d bench
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
MN_AB_LIB=build/ab/linkdbg.so python scripts/probe_link.py 300000 -1 > $O/link_dbg2.log 2>&1; cat $O/link_dbg2.log
python -m pytest tests/test_parallel.py tests/test_node2vec.py -m gpu -x -q > $O/t_call7a.log 2>&1; echo "parallel rc=$?"; tail -3 $O/t_call7a.log
python -m pytest tests/test_gpu_hnsw.py tests/test_schedule_pins.py -m gpu -x -q > $O/t_call7b.log 2>&1; echo "hnsw rc=$?"; tail -3 $O/t_call7b.log
python bench.py --dim 128 --no-wave-leg --recall-target 0 --quality-n 0 --no-graph-block --exact-inserts 200 --steps 10 > $O/bench_128b.json 2> $O/bench_128b.err; echo "bench128 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_128b.json"))
print("128-d:", d["value"], d["roofline"]["frac"], d["build_vectors_per_s"], d["build_roofline"]["frac"], d["parity_vs_oracle"]["vs_reference_binary"])
PY
