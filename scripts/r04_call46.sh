#!/bin/bash
# round-4 GPU call 46: the default bench.py run on the final tree (the graph block's traffic figures are stamped again)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
( time timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?"; cat $O/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["build_vectors_per_s"], d["build_roofline"]["frac"])
print(d["build_exact_at_full_size"], d["one_query_per_call"]["ms_per_query_median"])
for k,v in d["graph"].items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("traffic"), v.get("leg_wall_s"))
PY
timeout -k 10 200 python bench_graph.py --workload tvf --no-ref-sql > $O/tvf_bench_noref.json 2> $O/tvf_bench_noref.err; echo "tvf rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/tvf_bench_noref.json"):
    d=json.loads(l); print(d["metric"], d["config"]["device_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
