#!/bin/bash
# round-4 GPU call: Leiden parity + profiles, HNSW at 128-d (bench line + kernel stats)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
python -m pytest tests/test_leiden.py tests/test_fault_inject.py "tests/test_sqlite_ext.py::test_graph_leiden_sql_fast_mode_matches_oracle_schedule" "tests/test_sqlite_ext.py::test_rolled_back_savepoints_and_failed_statements_take_their_queued_rows_with_them" -m gpu -x -q > $O/t_leiden.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_leiden.log
python scripts/probe_leiden.py 3 > $O/lei_u.log 2>&1; cat $O/lei_u.log
python scripts/probe_leiden.py 3 500000 weighted > $O/lei_w.log 2>&1; cat $O/lei_w.log
bash scripts/prof_leiden.sh "" r04c | tail -2
bash scripts/prof_leiden.sh weighted r04cw | tail -2
cd "$R"
python bench.py --dim 128 --no-wave-leg --recall-target 0 --quality-n 0 --no-graph-block --exact-inserts 200 --steps 10 > $O/bench_128.json 2> $O/bench_128.err; echo "bench128 rc=$?"; tail -c 600 $O/bench_128.json
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_128_k -o b128 -- python3 $R/bench.py --dim 128 --no-wave-leg --recall-target 0 --quality-n 0 --no-graph-block --no-cpu-baseline --ef-sweep "" --steps 5 > $O/prof_r04_128_k.log 2>&1; echo "prof128 rc=$?"
