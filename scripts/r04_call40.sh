#!/bin/bash
# round-4 GPU call 40: Brandes with narrow workgroups and a scratch budget taken from the free memory (parity tests, bench leg,
# A/B against the old launch shape on the same box); f-2 timing beside the compiled reference (scripts/probe_csr_delta.py);
# f-4 kernel stats + FETCH/WRITE passes (scripts/prof_tvf.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 400 python -m pytest tests/test_graph_tvf.py tests/test_fault_inject.py -m gpu -x -q > $O/t_call40.log 2>&1; rc=$?; echo "tvf tests rc=$rc"; tail -3 $O/t_call40.log
[ $rc -eq 0 ] || exit $rc
for V in "new:" "old:MN_BRANDES_LANES=64 MN_BRANDES_SCRATCH_MB=8192" "lanes16:MN_BRANDES_LANES=16" "lanes64_one_chunk:MN_BRANDES_LANES=64"; do
  T=${V%%:*}; E=${V#*:}
  env $E timeout -k 10 200 python bench_graph.py --workload betweenness --no-ref-sql > $O/bc_$T.json 2> $O/bc_$T.err; echo "bc $T rc=$?"
  python -c "
import json,sys
d=json.loads(open('$O/bc_$T.json').read().strip().splitlines()[-1])
print('$T', d['config']['device_ms'], d['at_published_size_through_sql']['this_extension_ms'], d['roofline']['frac'])"
done
timeout -k 10 300 python bench_graph.py --workload betweenness > $O/tvf_betweenness.json 2> $O/tvf_betweenness.err; echo "bc full rc=$?"; cut -c1-900 $O/tvf_betweenness.json
timeout -k 10 200 python scripts/probe_csr_delta.py > $O/csr_delta.json 2> $O/csr_delta.err; echo "csr rc=$?"; cat $O/csr_delta.json; tail -3 $O/csr_delta.err
bash scripts/prof_tvf.sh
cd "$R"
KS=$(ls $O/prof_tvf_k/*kernel_stats.csv $O/prof_tvf_k/*/*kernel_stats.csv 2>/dev/null | head -1)
FC=$(ls $O/prof_tvf_f/*counter_collection.csv $O/prof_tvf_f/*/*counter_collection.csv 2>/dev/null | head -1)
WC=$(ls $O/prof_tvf_w/*counter_collection.csv $O/prof_tvf_w/*/*counter_collection.csv 2>/dev/null | head -1)
echo "$KS $FC $WC"
cp "$FC" $O/tvf_fetch_counters.csv; cp "$WC" $O/tvf_write_counters.csv
python scripts/summarize_prof.py r04_tvf_1M_20M "$KS" "$FC" "$WC" && cp profiles/r04_tvf_1M_20M_* $O/ && cat profiles/r04_tvf_1M_20M_pmc_summary.csv
