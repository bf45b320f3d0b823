#!/bin/bash
# round-4 GPU call 15 (final evidence, part 1): PMC + kernel-trace passes of the headline bench and of Leiden on the final kernels,
# randomised parity sweeps (HNSW; graph half incl. the synchronous Leiden schedule)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
bash scripts/prof_bench.sh r04_bench_1Mx768_sse | tail -3
bash scripts/prof_leiden.sh "" r04u | tail -2
bash scripts/prof_leiden.sh weighted r04w | tail -2
cd "$R"
python scripts/fuzz_parity.py 100 9191 > $O/fuzz_parity.log 2>&1; echo "fuzz hnsw rc=$?"; tail -2 $O/fuzz_parity.log
python scripts/fuzz_graph.py 100 9292 > $O/fuzz_graph.log 2>&1; echo "fuzz graph rc=$?"; tail -2 $O/fuzz_graph.log
