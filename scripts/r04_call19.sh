#!/bin/bash
# round-4 GPU call 19: where an expansion's time goes with the sorted register queue (phase-timing build), then the bulk-build
# reservation A/B (one allocation of the slot tables vs geometric regrowth), interleaved on one box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
for S in "3000 128 l2" "10000 128 l2" "10000 768 l2" "1000000 768 cosine"; do
  timeout -k 10 300 python scripts/probe_phases.py $S 2>&1 | tail -4
done > $O/phases_r04.log 2>&1
cat $O/phases_r04.log
for V in grow hint grow hint grow hint; do
  echo "== $V"
  if [ "$V" = grow ]; then MN_BUILD_NO_RESERVE=1 timeout -k 10 300 python scripts/probe_search_only.py sse 2>&1 | tail -3
  else timeout -k 10 300 python scripts/probe_search_only.py sse 2>&1 | tail -3; fi
done > $O/ab_reserve.log 2>&1
cat $O/ab_reserve.log
