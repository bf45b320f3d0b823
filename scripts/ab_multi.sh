#!/bin/bash
# usage: ab_multi.sh <order> <variant.so>...   (each variant timed on the same box, base first)
ORDER=$1; shift
cp sqlite-muninn_amd/libmuninn_hip.so /tmp/lib_base.so
echo "== base"; python scripts/probe_search_only.py $ORDER 2>&1 | tail -2
for V in "$@"; do
  cp build/ab/$V sqlite-muninn_amd/libmuninn_hip.so
  echo "== $V"; python scripts/probe_search_only.py $ORDER 2>&1 | tail -2
done
cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so
