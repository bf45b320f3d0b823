#!/bin/bash
# A/B on ONE box: the default library, then each build/ab/<variant>.so, single-query latency (scripts/probe_latency.py)
set -e
python scripts/probe_latency.py 2>&1 | grep "single query"
cp sqlite-muninn_amd/libmuninn_hip.so /tmp/lib_base.so
for VAR in "$@"; do
  echo "== $VAR"
  cp build/ab/$VAR sqlite-muninn_amd/libmuninn_hip.so
  python scripts/probe_latency.py 2>&1 | grep "single query"
done
cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so
