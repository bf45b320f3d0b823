#!/bin/bash
# A/B on ONE box: the default library, then each build/ab/<variant>.so, same search workload (1M x 768, ef 128 / 256)
set -e
ORDER=${1:-sse}; shift
python scripts/probe_search_only.py $ORDER 2>&1 | tail -3
cp sqlite-muninn_amd/libmuninn_hip.so /tmp/lib_base.so
for VAR in "$@"; do
  echo "== $VAR"
  cp build/ab/$VAR sqlite-muninn_amd/libmuninn_hip.so
  python scripts/probe_search_only.py $ORDER 2>&1 | tail -3
done
cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so
