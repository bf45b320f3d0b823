#!/bin/bash
# round-4 GPU call 31: link decisions of a window taken ahead of its commit (k_spec_prepare): the insert tests, then full-size
# exact inserts against the compiled reference with the window trace, prepare on / off
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q -k "specul or golden or sequential or tie or insert or exact or duplicate or m_above or overgrown" > $O/t_call31.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -3 $O/t_call31.log
[ $rc -eq 0 ] || exit 1
for A in 1 0; do
  MN_SPEC_AHEAD=$A MN_SPEC_TRACE=1 timeout -k 10 500 python bench.py --no-graph-block --no-wave-leg --quality-n 0 --recall-target 0 --ef-sweep "" --exact-inserts 2000 --steps 5 --ref-queries 200 --cpu-queries 200 > $O/bench_exact_$A.json 2> $O/bench_exact_$A.err; echo "bench ahead=$A rc=$?"
  grep "mn_spec" $O/bench_exact_$A.err | tail -1
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_exact_$A.json"))
print(d["build_exact_at_full_size"])
PY
done
for A in 0 1 0 1; do
  echo "== MN_SPEC_AHEAD=$A"
  MN_SPEC_AHEAD=$A timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -3
done > $O/ab_ahead.log 2>&1
cat $O/ab_ahead.log
