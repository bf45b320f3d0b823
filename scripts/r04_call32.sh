#!/bin/bash
# round-4 GPU call 32 (final evidence, part 1): counter passes of the headline bench on the final kernel sources; the SQL surface
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
bash scripts/prof_bench.sh r04_bench_1Mx768_sse | tail -4
cd "$R"
for f in k f w; do ls $O/prof_r04_bench_1Mx768_sse_$f/*/ 2>/dev/null | head -3; done
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
