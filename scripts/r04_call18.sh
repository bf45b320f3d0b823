#!/bin/bash
# round-4 GPU call 18: the sorted single-array register queue (beam_layer_regs): parity suite, then latency against the
# previous library (build/ab/queue_v1.so = the tree before this change), interleaved on one box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q > $O/t_call18.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -5 $O/t_call18.log
[ $rc -eq 0 ] || exit 1
for V in v1 v2 v1 v2; do
  echo "== $V"
  if [ "$V" = v1 ]; then MN_AB_LIB=build/ab/queue_v1.so timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4
  else timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4; fi
done > $O/ab_queue.log 2>&1
cat $O/ab_queue.log
for V in v1 v2; do
  echo "== $V 1M"
  if [ "$V" = v1 ]; then MN_AB_LIB=build/ab/queue_v1.so timeout -k 10 400 python scripts/probe_latency3.py 2>&1 | tail -1
  else timeout -k 10 400 python scripts/probe_latency3.py 2>&1 | tail -1; fi
done > $O/ab_queue_1m.log 2>&1
cat $O/ab_queue_1m.log
