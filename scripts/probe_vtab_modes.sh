#!/bin/bash
# SQL-surface insert throughput in the three MUNINN_HNSW_MODE settings (run on the GPU box)
set -e
cd "$(dirname "$0")/.."
for m in exact deferred; do echo "== $m"; MUNINN_HNSW_MODE=$m timeout -k 10 300 python -u scripts/probe_vtab.py 10000 2>&1 | tail -4; done
echo "== fast (100k rows)"; MUNINN_HNSW_MODE=fast timeout -k 10 300 python -u scripts/probe_vtab.py 100000 2>&1 | tail -4
