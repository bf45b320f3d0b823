#!/bin/bash
# round-4 GPU call 49: FETCH_SIZE / WRITE_SIZE passes of the headline workload at 128 floats per row (config 1's / config 4's row width)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -d "$R/sqlite-muninn_amd" ] || exit 1
O=$R/gpurun_out
mkdir -p "$O"
ARGS="--dim 128 --steps 5 --warmup 1 --no-cpu-baseline --no-wave-leg --exact-inserts 0 --quality-n 0 --recall-target 0 --ef-sweep , --recall-queries 100"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 140 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_128_f -o b -- python3 $R/bench.py $ARGS > $O/prof_128_f.log 2>&1; echo "f rc=$?"
timeout -k 5 140 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_128_w -o b -- python3 $R/bench.py $ARGS > $O/prof_128_w.log 2>&1; echo "w rc=$?"
cd "$R"
python - <<'PY'
import csv, glob, statistics as st
for p, c in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/prof_128_{p}/*counter_collection.csv")
    if not f:
        print(p, "no counter file"); continue
    by = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and "k_beam" in r["Kernel_Name"]:
            by.setdefault(r["Kernel_Name"], []).append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    for k, v in by.items():
        print(c, k[:60], len(v), round(st.mean(x[0] for x in v), 1), round(st.mean(x[1] for x in v), 3))
PY
ls -la $O/prof_128_f $O/prof_128_w | head; du -sh $O/prof_128_f $O/prof_128_w
