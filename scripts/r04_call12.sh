#!/bin/bash
# round-4 GPU call 12: speculative row requests in the latency kernels (parity + same-process A/B), node2vec multi-value reduction A/B
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
python -m pytest tests/test_gpu_hnsw.py tests/test_sqlite_ext.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/t_call12.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_call12.log
python scripts/probe_latency2.py > $O/lat2.log 2>&1; cat $O/lat2.log
bash scripts/ab_n2v.sh n2vmulti.so > $O/ab_n2v2.log 2>&1; cat $O/ab_n2v2.log
MN_AB_LIB=x python - <<'PY'
PY
