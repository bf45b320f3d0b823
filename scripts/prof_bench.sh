#!/bin/bash
set -u
# rocprofv3 passes over the headline workload (bench.py, 1M x 768, 10k queries, k=10, ef=128; legs that are not the timed
# kernel switched off): kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs.  usage: prof_bench.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$R/sqlite-muninn_amd" ] || { echo "repository root not found: $R" >&2; exit 1; }
T=${1:-r03_bench_1Mx768_sse}
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-wave-leg --exact-inserts 0 --quality-n 0 --recall-target 0 --ef-sweep , --recall-queries 100"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_k -o b -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${T}_k.log 2>&1; echo "k rc=$?"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_f -o b -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${T}_f.log 2>&1; echo "f rc=$?"
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_w -o b -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${T}_w.log 2>&1; echo "w rc=$?"
ls $R/gpurun_out/prof_${T}_*/ | head -20
