#!/bin/bash
# round-4 GPU call 22: speculative expansion of the runner-up (pair path of beam_layer_regs): parity suite, then latency with
# the path off (MN_PAIR=0) and on, interleaved, same library, same box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q > $O/t_call22.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -5 $O/t_call22.log
[ $rc -eq 0 ] || exit 1
for V in 0 1 0 1; do
  echo "== MN_PAIR=$V"
  MN_PAIR=$V timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4
done > $O/ab_pair.log 2>&1
cat $O/ab_pair.log
