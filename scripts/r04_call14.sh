#!/bin/bash
# round-4 GPU call 14: the build's link half divided over ranks (2- and 4-rank parity, failure behaviour), whole parallel suite
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
python -m pytest tests/test_parallel.py -m gpu -x -q > $O/t_call14.log 2>&1; echo "pytest rc=$?"; tail -4 $O/t_call14.log
