#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$R/sqlite-muninn_amd" ] || { echo "repository root not found: $R" >&2; exit 1; }
cd $R
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import muninn_amd
from bench_graph import er_edges
pkg = muninn_amd.pkg
for n, m, path in ((300, 3000, "/tmp/n2v_tiny.csr"), (4000, 80000, "/tmp/n2v_4k.csr")):
    off, adj = pkg.graph.n2v_csr_from_edges(n, *er_edges(n, m))
    with open(path, "wb") as f:
        f.write(np.int32(n).tobytes() + np.int64(len(adj)).tobytes() + np.ascontiguousarray(off, np.int32).tobytes() + np.ascontiguousarray(adj, np.int32).tobytes())
PY
g++ -O2 -o /tmp/n2v_bench tools/n2v_bench.cpp -Iinclude -Lsqlite-muninn_amd -lmuninn_hip -Wl,-rpath,$R/sqlite-muninn_amd
cd /tmp && export TMPDIR=/tmp
try() { name=$1; shift; echo "== $name"; "$@" > $R/gpurun_out/try_$name.log 2>&1; echo "rc=$?"; grep -c "SIGSEGV" $R/gpurun_out/try_$name.log; grep "^{" $R/gpurun_out/try_$name.log | cut -c1-160; }
try tiny_seq rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/try_tiny_seq -o n2v -- /tmp/n2v_bench /tmp/n2v_tiny.csr 1 0 seq
try tiny_bat rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/try_tiny_bat -o n2v -- /tmp/n2v_bench /tmp/n2v_tiny.csr 1
try k4_bat rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/try_k4_bat -o n2v -- /tmp/n2v_bench /tmp/n2v_4k.csr 1
try k4_sq rocprofv3 --pmc SQ_WAVES --output-format csv -d $R/gpurun_out/try_k4_sq -o n2v -- /tmp/n2v_bench /tmp/n2v_4k.csr 1
try k4_v2 rocprofv2 --pmc FETCH_SIZE -d $R/gpurun_out/try_k4_v2 /tmp/n2v_bench /tmp/n2v_4k.csr 1
ls -R $R/gpurun_out/try_*/ 2>/dev/null | head -30
