#!/bin/bash
# round-4 GPU call 13: weighted Leiden evaluation by community slots (parity, timing), speculative rows gated by index size
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
python -m pytest tests/test_leiden.py "tests/test_parallel.py::test_leiden_divided_over_ranks_is_bit_identical_to_one_gpu" tests/test_fault_inject.py -m gpu -x -q > $O/t_call13.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_call13.log
python scripts/probe_leiden.py 3 500000 weighted > $O/lei_w.log 2>&1; cat $O/lei_w.log
python scripts/probe_leiden.py 2 > $O/lei_u.log 2>&1; cat $O/lei_u.log
python scripts/fuzz_graph.py 120 8484 > $O/fuzz_graph.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz_graph.log
