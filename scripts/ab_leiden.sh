#!/bin/bash
# A/B on ONE box: default library then build/ab/<variant>.so on bench_graph's Leiden workload
run() { python bench_graph.py --workload leiden --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), j['modularity'], j['sweeps'], j['parity_vs_oracle']['communities_identical'])"; }
echo "== default"; run; MN_LEIDEN_SG=32 run
cp sqlite-muninn_amd/libmuninn_hip.so /tmp/lib_base.so
for VAR in "$@"; do echo "== $VAR"; cp build/ab/$VAR sqlite-muninn_amd/libmuninn_hip.so; run; MN_LEIDEN_SG=32 run; done
cp /tmp/lib_base.so sqlite-muninn_amd/libmuninn_hip.so
