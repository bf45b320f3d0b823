#!/bin/bash
# round-4 GPU call 45: Brandes launch shapes in the parity test; Leiden counter passes again (mn_graph.hip changed in its Brandes
# section, which made the file-level stamp of the Leiden traffic entries stale)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 400 python -m pytest tests/test_graph_tvf.py tests/test_leiden.py -m gpu -x -q > $O/t_call45.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/t_call45.log
[ $rc -eq 0 ] || exit $rc
bash scripts/prof_leiden.sh "" r04u2 | tail -3
bash scripts/prof_leiden.sh weighted r04w2 | tail -3
cd "$R"
for T in u w; do
  N=unweighted; [ $T = w ] && N=weighted
  KS=$(ls $O/prof_r04${T}2_k/*kernel_stats.csv | head -1); FC=$(ls $O/prof_r04${T}2_f/*counter_collection.csv | head -1); WC=$(ls $O/prof_r04${T}2_w/*counter_collection.csv | head -1)
  python scripts/summarize_prof.py r04_leiden_500k_9M_$N "$KS" "$FC" "$WC" && cp profiles/r04_leiden_500k_9M_${N}_* $O/
done
