#!/bin/bash
# round-4 GPU call 29: fine-grained validation of speculative windows (+ trimmed merge loop): parity suite, full-size exact
# inserts against the compiled reference with the window trace, then the lone-query A/B against walkplain.so
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q > $O/t_call29.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -3 $O/t_call29.log
[ $rc -eq 0 ] || exit 1
MN_SPEC_TRACE=1 timeout -k 10 500 python bench.py --no-graph-block --no-wave-leg --quality-n 0 --recall-target 0 --ef-sweep "" --exact-inserts 2000 --steps 5 --ref-queries 200 --cpu-queries 200 > $O/bench_exact.json 2> $O/bench_exact.err; echo "bench rc=$?"
grep "mn_spec" $O/bench_exact.err | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_exact.json"))
print(d["value"], d["build_exact_at_full_size"], d.get("one_query_per_call"))
PY
for V in walkplain product walkplain product; do
  echo "== $V"
  if [ "$V" = product ]; then timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -5
  else MN_AB_LIB=build/ab/$V.so timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -5; fi
done > $O/ab_trim.log 2>&1
cat $O/ab_trim.log
