#!/bin/bash
# round-4 GPU call 34-35: greedy steps re-taken on rewritten rows, heap-path searches logged too (speculative windows): insert tests, full-size exact inserts against
# the compiled reference with the window trace, small sizes
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q > $O/t_call34.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -3 $O/t_call34.log
[ $rc -eq 0 ] || exit 1
MN_SPEC_TRACE=1 timeout -k 10 500 python bench.py --no-graph-block --no-wave-leg --quality-n 0 --recall-target 0 --ef-sweep "" --exact-inserts 3000 --steps 5 --ref-queries 200 --cpu-queries 200 > $O/bench_exact_g.json 2> $O/bench_exact_g.err; echo "bench rc=$?"
grep "mn_spec" $O/bench_exact_g.err | tail -2
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_exact_g.json"))
print(d["build_exact_at_full_size"])
PY
MN_SPEC_TRACE=1 timeout -k 10 300 python scripts/probe_latency3.py small > $O/lat_g.log 2>&1; grep -v "^\[mn_spec\] windows" $O/lat_g.log | tail -8 | cut -c1-250
