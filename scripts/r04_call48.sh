#!/bin/bash
# round-4 GPU call 48: Leiden counter passes on the final mn_graph.hip (the Brandes scratch budget now leaves the edge matrix its room;
# the file-level stamp of the Leiden traffic entries follows the file)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
rm -rf $O/prof_r04u2_* $O/prof_r04w2_*
bash scripts/prof_leiden.sh "" r04u2 | tail -3
bash scripts/prof_leiden.sh weighted r04w2 | tail -3
cd "$R"
for T in u w; do
  N=unweighted; [ $T = w ] && N=weighted
  KS=$(ls $O/prof_r04${T}2_k/*kernel_stats.csv | head -1); FC=$(ls $O/prof_r04${T}2_f/*counter_collection.csv | head -1); WC=$(ls $O/prof_r04${T}2_w/*counter_collection.csv | head -1)
  python scripts/summarize_prof.py r04_leiden_500k_9M_$N "$KS" "$FC" "$WC" && cp profiles/r04_leiden_500k_9M_${N}_* $O/
done
