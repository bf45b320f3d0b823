#!/bin/bash
# round-4 GPU call 17: does ONE allocation of the slot tables (bulk-build hint) change the search kernel's run-to-run spread?
# interleaved trials of the same library: geometric regrowth (MN_BUILD_NO_RESERVE=1) vs the hint (default)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
for V in grow hint grow hint grow hint grow hint; do
  echo "== $V"
  if [ "$V" = grow ]; then MN_BUILD_NO_RESERVE=1 python scripts/probe_search_only.py sse 2>&1 | tail -3
  else python scripts/probe_search_only.py sse 2>&1 | tail -3; fi
done > $O/ab_reserve.log 2>&1
cat $O/ab_reserve.log
