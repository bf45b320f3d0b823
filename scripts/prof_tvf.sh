#!/bin/bash
set -u
# round-4 f-4 profiles: kernel trace + separate PMC passes (FETCH_SIZE, WRITE_SIZE) over bench_graph.py's three TVF workloads
# at their C-ABI sizes (pagerank / components 1M nodes, 20M rows; Brandes 20k nodes), the reference's SQL run skipped
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$R/sqlite-muninn_amd" ] || { echo "repository root not found: $R" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
for P in k f w; do
  case $P in k) OPT="--kernel-trace --stats";; f) OPT="--pmc FETCH_SIZE";; w) OPT="--pmc WRITE_SIZE";; esac
  timeout -k 5 280 rocprofv3 $OPT --output-format csv -d $R/gpurun_out/prof_tvf_$P -o tvf -- python3 $R/bench_graph.py --workload tvf --no-ref-sql > $R/gpurun_out/prof_tvf_$P.log 2>&1; echo "$P rc=$?"
done
cut -c1-300 $R/gpurun_out/prof_tvf_k.log | tail -4
