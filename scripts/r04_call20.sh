#!/bin/bash
# round-4 GPU call 20: the SQL surface like for like (reference extension on ALL the same rows) with the sorted register queue,
# and how often the next pop is the previous pop's runner-up (would a speculative second expansion be used?)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
timeout -k 10 500 python bench_sql.py --n 10000 --dim 128 --ref-n 10000 > $O/sql_10kx128.json 2> $O/sql_10kx128.err; echo "sql1 rc=$?"
timeout -k 10 500 python bench_sql.py --n 10000 --dim 768 --ref-n 10000 > $O/sql_10kx768.json 2> $O/sql_10kx768.err; echo "sql2 rc=$?"
timeout -k 10 300 python bench_sql.py --n 3000 --dim 128 --ref-n 3000 > $O/sql_3kx128.json 2> $O/sql_3kx128.err; echo "sql3 rc=$?"
python - <<'PY'
import json
for f in ("sql_10kx128", "sql_10kx768", "sql_3kx128"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, {m: {k: v for k, v in d["modes"][m].items() if "per" in k or "ms" in k} for m in d["modes"]}, d["cpu_baseline"])
    except Exception as e:
        print(f, "unreadable", e)
PY
for S in "3000 128 l2" "10000 128 l2" "10000 768 l2" "1000000 768 cosine"; do
  timeout -k 10 300 python scripts/probe_phases.py $S 2>&1 | tail -3
done > $O/phases_r04b.log 2>&1
cat $O/phases_r04b.log
