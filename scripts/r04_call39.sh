#!/bin/bash
# round-4 GPU call 39 (re-entry after the container was re-created): the whole GPU suite on the rebuilt tree (without -x: every
# failure is listed), smoke(), then the default bench.py run as the driver runs it
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
mkdir -p "$O"
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > $O/t_all.log 2>&1; rc=$?; echo "all rc=$rc"; tail -25 $O/t_all.log
[ $rc -le 1 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?"; cat $O/bench_default.time
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["build_vectors_per_s"], d["build_roofline"]["frac"])
print(d["build_exact_at_full_size"], d["one_query_per_call"]["ms_per_query_median"])
for k,v in d["graph"].items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("roofline",{}).get("traffic"), v.get("leg_wall_s"))
PY
