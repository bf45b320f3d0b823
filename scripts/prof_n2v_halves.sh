#!/bin/bash
set -u
# per-half PMC passes of config 4 (tools/n2v_kernels.hip); every profiler run is bounded: rocprofv3's counter collection has
# hung on this pipeline before
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$R/sqlite-muninn_amd" ] || { echo "repository root not found: $R" >&2; exit 1; }
cd $R
python bench_graph.py --workload node2vec --dump-csr /tmp/n2v.csr --dump-only
hipcc --offload-arch=gfx950 -O2 -o /tmp/n2v_kernels tools/n2v_kernels.hip -Iinclude -Lsqlite-muninn_amd -lmuninn_hip -Wl,-rpath,$R/sqlite-muninn_amd 2>/dev/null
cd /tmp && export TMPDIR=/tmp
try() { name=$1; shift; echo "== $name"; timeout -k 5 90 "$@" > $R/gpurun_out/h_$name.log 2>&1; echo "rc=$?"; grep "^{" $R/gpurun_out/h_$name.log | cut -c1-160; }
try plain /tmp/n2v_kernels /tmp/n2v.csr both 4
try samples_f rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/h_samples_f -o n2v -- /tmp/n2v_kernels /tmp/n2v.csr samples 4
try samples_w rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/h_samples_w -o n2v -- /tmp/n2v_kernels /tmp/n2v.csr samples 4
try both_f rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/h_both_f -o n2v -- /tmp/n2v_kernels /tmp/n2v.csr both 4
try both_w rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/h_both_w -o n2v -- /tmp/n2v_kernels /tmp/n2v.csr both 4
ls -la $R/gpurun_out/h_*/ 2>/dev/null | head -30
