#!/bin/bash
# round-4 GPU call 23: phase timers of the pair path (MN_PAIR=0 / 1)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
for P in 0 1; do
for S in "3000 128 l2" "10000 128 l2"; do
  echo "== MN_PAIR=$P $S"
  MN_PAIR=$P timeout -k 10 300 python scripts/probe_phases.py $S 2>&1 | tail -3
done
done > $O/phases_pair.log 2>&1
cat $O/phases_pair.log
