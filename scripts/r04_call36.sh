#!/bin/bash
# round-4 GPU call 36 (final evidence on the final kernels, part 1): insert tests with windows of up to 128, the window trace at full
# size against the compiled reference, counter passes of the headline bench, the SQL surface
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_hnsw.py -m gpu -x -q -k "specul or golden or sequential or insert or exact or fallback" > $O/t_call36.log 2>&1; rc=$?; echo "hnsw rc=$rc"; tail -2 $O/t_call36.log
[ $rc -eq 0 ] || exit 1
MN_SPEC_TRACE=1 timeout -k 10 500 python bench.py --no-graph-block --no-wave-leg --quality-n 0 --recall-target 0 --ef-sweep "" --exact-inserts 3000 --steps 5 --ref-queries 200 --cpu-queries 200 > $O/bench_exact_trace.json 2> $O/bench_exact_trace.err; echo "trace rc=$?"
grep "mn_spec" $O/bench_exact_trace.err | tail -2
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_exact_trace.json"))
print(d["build_exact_at_full_size"])
PY
bash scripts/prof_bench.sh r04_bench_1Mx768_sse | tail -4
cd "$R"
timeout -k 10 500 python bench_sql.py --n 10000 --dim 128 --ref-n 10000 > $O/sql_10kx128.json 2> $O/sql_10kx128.err; echo "sql1 rc=$?"
timeout -k 10 500 python bench_sql.py --n 10000 --dim 768 --ref-n 10000 > $O/sql_10kx768.json 2> $O/sql_10kx768.err; echo "sql2 rc=$?"
timeout -k 10 300 python bench_sql.py --n 3000 --dim 128 --ref-n 3000 > $O/sql_3kx128.json 2> $O/sql_3kx128.err; echo "sql3 rc=$?"
python - <<'PY'
import json
for f in ("sql_10kx128", "sql_10kx768", "sql_3kx128"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, {m: {k: round(v, 4) for k, v in d["modes"][m].items() if "per" in k or "ms" in k or "rate" in k} for m in d["modes"]},
              {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["cpu_baseline"].items() if k != "sample"})
    except Exception as e:
        print(f, "unreadable", e)
PY
