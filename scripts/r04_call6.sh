#!/bin/bash
# round-4 GPU call 6: link-step counters on real node2vec embeddings, Leiden divided over ranks (parity), quad loads: interleaved trials
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
MN_AB_LIB=build/ab/linkdbg.so python scripts/probe_link.py 300000 -1 > $O/link_dbg2.log 2>&1; cat $O/link_dbg2.log
python -m pytest tests/test_leiden.py "tests/test_parallel.py::test_leiden_divided_over_ranks_is_bit_identical_to_one_gpu" -m gpu -x -q > $O/t_call6.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_call6.log
cp sqlite-muninn_amd/libmuninn_hip.so /tmp/lib_base.so
for V in base quad12 quad16 base quad12 quad16 base quad12; do
  if [ "$V" = base ]; then cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so; else cp build/ab/$V.so sqlite-muninn_amd/libmuninn_hip.so; fi
  echo "== $V"; python scripts/probe_search_only.py sse 2>&1 | tail -2
done > $O/ab_quad4.log 2>&1
cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so
cat $O/ab_quad4.log
