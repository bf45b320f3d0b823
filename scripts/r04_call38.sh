#!/bin/bash
# round-4 GPU call 38: the randomized speculative-window test (new) and its neighbours
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
MN_SPEC_TRACE=1 timeout -k 10 600 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q -k "specul" -s > $O/t_call38.log 2>&1; rc=$?; echo "rc=$rc"; grep -c "mn_spec\] windows" $O/t_call38.log; grep "mn_spec\] [0-9]" $O/t_call38.log | tail -14 | cut -c1-200; tail -3 $O/t_call38.log
