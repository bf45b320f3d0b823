#!/bin/bash
# round-4 GPU call 26: leader-less distance requests + merge without compaction: parity suite, then A/B against
# build/ab/nopair.so (the sorted queue alone), interleaved
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q > $O/t_call26.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -3 $O/t_call26.log
[ $rc -eq 0 ] || exit 1
for V in v1 v2 v1 v2; do
  echo "== $V"
  if [ "$V" = v1 ]; then MN_AB_LIB=build/ab/nopair.so timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4
  else timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4; fi
done > $O/ab_alone.log 2>&1
cat $O/ab_alone.log
