#!/bin/bash
# round-4 GPU call 2: Leiden after the refinement skip (parity, timing, SQ counters), Node2Vec -> index leg under the kernel
# trace, and the default bench.py run with its graph block
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
python -m pytest tests/test_leiden.py -m gpu -x -q > $O/t_leiden.log 2>&1; echo "pytest rc=$?"; tail -2 $O/t_leiden.log
python scripts/probe_leiden.py 3 > $O/lei_u.log 2>&1; cat $O/lei_u.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/prof_r04d_sq -o lei -- python3 $R/scripts/probe_leiden.py 1 500000 > $O/prof_r04d_sq.log 2>&1; echo "sq rc=$?"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_n2v_k -o n2v -- python3 $R/bench_graph.py --workload node2vec --steps 1 --warmup 0 > $O/n2v_bench.json 2> $O/n2v_bench.err; echo "n2v rc=$?"; tail -c 1500 $O/n2v_bench.json
cd "$R"
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?"; cat $O/bench_default.time; tail -c 800 $O/bench_default.json
