#!/bin/bash
# usage: ab_n2v.sh <variant.so>...  — node2vec batched timing of lib variants (build/ab/) on the same box, base first
cp sqlite-muninn_amd/libmuninn_hip.so /tmp/lib_base.so
echo "== base"; python scripts/probe_n2v.py 1000000 20000000 128 1 80 2>&1 | tail -1
for V in "$@"; do
  cp build/ab/$V sqlite-muninn_amd/libmuninn_hip.so
  echo "== $V"; python scripts/probe_n2v.py 1000000 20000000 128 1 80 2>&1 | tail -1
done
cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so
