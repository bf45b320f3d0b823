#!/bin/bash
# round-4 GPU call 27: leader-less requests + merge without compaction + lean lone-query stream (+ prefetching tile walk):
# parity suite, then three libraries interleaved: nopair (sorted queue alone), walkplain (all but the walk), product
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q > $O/t_call27.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -3 $O/t_call27.log
[ $rc -eq 0 ] || exit 1
for V in nopair walkplain product nopair walkplain product; do
  echo "== $V"
  if [ "$V" = product ]; then timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4
  else MN_AB_LIB=build/ab/$V.so timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4; fi
done > $O/ab_alone.log 2>&1
cat $O/ab_alone.log
