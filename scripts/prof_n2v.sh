#!/bin/bash
set -u
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$R/sqlite-muninn_amd" ] || { echo "repository root not found: $R" >&2; exit 1; }
cd $R
python bench_graph.py --workload node2vec --dump-csr /tmp/n2v.csr --dump-only
g++ -O2 -o /tmp/n2v_bench tools/n2v_bench.cpp -Iinclude -Lsqlite-muninn_amd -lmuninn_hip -Wl,-rpath,$R/sqlite-muninn_amd
/tmp/n2v_bench /tmp/n2v.csr 2 | tee gpurun_out/n2v_plain.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_n2v_k -o n2v -- /tmp/n2v_bench /tmp/n2v.csr 1 > $R/gpurun_out/prof_n2v_k.log 2>&1
echo kernel-trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_n2v_f -o n2v -- /tmp/n2v_bench /tmp/n2v.csr 1 > $R/gpurun_out/prof_n2v_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_n2v_w -o n2v -- /tmp/n2v_bench /tmp/n2v.csr 1 > $R/gpurun_out/prof_n2v_w.log 2>&1
echo write done
ls -la $R/gpurun_out/prof_n2v_*/
grep "^{" $R/gpurun_out/prof_n2v_f.log $R/gpurun_out/prof_n2v_w.log $R/gpurun_out/prof_n2v_k.log
