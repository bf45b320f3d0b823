#!/bin/bash
# round-4 GPU call 5: link-step counters on concentrated data, quad loads A/B (more in flight), walk_grad draw-ahead A/B
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
MN_AB_LIB=build/ab/linkdbg.so python scripts/probe_link.py 300000 0.6 > $O/link_dbg.log 2>&1; cat $O/link_dbg.log
MN_AB_LIB=build/ab/linkdbg.so python scripts/probe_link.py 300000 0.0 >> $O/link_dbg.log 2>&1; tail -2 $O/link_dbg.log
bash scripts/ab_search2.sh sse quad12.so quad16.so quad24.so quad0.so quad12.so > $O/ab_quad3.log 2>&1; cat $O/ab_quad3.log
bash scripts/ab_n2v.sh n2vdraw.so > $O/ab_n2v.log 2>&1; cat $O/ab_n2v.log
