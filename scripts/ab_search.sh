#!/bin/bash
# A/B: default lib vs build/ab/<variant>.so on the same box (search kernel, 1M x 768)
set -e
ORDER=${1:-sse}; VAR=${2:-libmuninn_c256.so}
python scripts/probe_search_only.py $ORDER 2>&1 | tail -2
cp sqlite-muninn_amd/libmuninn_hip.so /tmp/lib_base.so
cp build/ab/$VAR sqlite-muninn_amd/libmuninn_hip.so
python scripts/probe_search_only.py $ORDER 2>&1 | tail -2
cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so
