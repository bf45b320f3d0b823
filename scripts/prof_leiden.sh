#!/bin/bash
set -u
# round-3 Leiden profiles: kernel trace + separate PMC passes over ONE run_leiden on the config-5 graph (scripts/probe_leiden.py 1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$R/sqlite-muninn_amd" ] || { echo "repository root not found: $R" >&2; exit 1; }
cd /tmp && export TMPDIR=/tmp
W=${1:-}
T=${2:-lei}
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_k -o lei -- python3 $R/scripts/probe_leiden.py 1 500000 $W > $R/gpurun_out/prof_${T}_k.log 2>&1; echo "k rc=$?"
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_f -o lei -- python3 $R/scripts/probe_leiden.py 1 500000 $W > $R/gpurun_out/prof_${T}_f.log 2>&1; echo "f rc=$?"
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_w -o lei -- python3 $R/scripts/probe_leiden.py 1 500000 $W > $R/gpurun_out/prof_${T}_w.log 2>&1; echo "w rc=$?"
grep "^run" $R/gpurun_out/prof_${T}_k.log $R/gpurun_out/prof_${T}_f.log
