#!/bin/bash
# round-4 GPU call 25: (a) is the pair path worth its code? product library (pair compiled in) vs build/ab/nopair.so, interleaved;
# (b) instruction-cache counters of the lone-query kernel
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out
for V in nopair pair nopair pair; do
  echo "== $V"
  if [ "$V" = nopair ]; then MN_AB_LIB=build/ab/nopair.so timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4
  else timeout -k 10 300 python scripts/probe_latency3.py small 2>&1 | tail -4; fi
done > $O/ab_pair_code.log 2>&1
cat $O/ab_pair_code.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|SQ_INSTS_VALU |SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INST_CYCLES|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_IFETCH" | cut -c1-160 | sort -u | head -30 > $O/pmc_avail.log 2>&1
cat $O/pmc_avail.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $O/prof_icache -o ic --output-format csv -- python3 $R/scripts/probe_lone.py > $O/prof_icache.log 2>&1; echo "pmc1 rc=$?"; tail -2 $O/prof_icache.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O/prof_sq -o sq --output-format csv -- python3 $R/scripts/probe_lone.py > $O/prof_sq.log 2>&1; echo "pmc2 rc=$?"; tail -2 $O/prof_sq.log
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("prof_icache", "prof_sq"):
    for f in glob.glob(f"gpurun_out/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k in acc:
            if "coop" in k or "beam" in k:
                print(tag, k, {c: round(v / max(cnt[(k, c)], 1)) for c, v in acc[k].items()}, "dispatches", max(cnt[(k, c)] for c in acc[k]))
PY
