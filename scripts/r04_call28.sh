#!/bin/bash
# round-4 GPU call 28: trimmed merge loop (product) vs walkplain.so (same tree before the trim), interleaved; lone tests first
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q -k "lone or latency or golden or tie or sequential or duplicate" > $O/t_call28.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -3 $O/t_call28.log
[ $rc -eq 0 ] || exit 1
for V in walkplain product walkplain product; do
  echo "== $V"
  if [ "$V" = product ]; then timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -5
  else MN_AB_LIB=build/ab/$V.so timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -5; fi
done > $O/ab_trim.log 2>&1
cat $O/ab_trim.log
