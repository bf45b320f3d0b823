"""Re-stamp the two k_beam entries of profiles/traffic.json from profiles/<tag>_pmc_summary.csv (scripts/prof_bench.sh +
scripts/summarize_prof.py): mean FETCH_SIZE / WRITE_SIZE per launch of the search and the build instantiation, FETCH_SIZE doubled
(gfx950: the counter counts 32-byte units where its description says 64, MI355X_MICROARCH.md), and the sha-256 of the kernel's sources
as bench.py computes it.  usage: stamp_traffic_bench.py <tag> <commit> [kernel avg ms]"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha
tag, commit = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv"))))
def pick(counter, build):
    pat = re.compile(r"k_beam<0, \d, %s, false>\(MnDevIndex, MnSearchArgs\)" % ("true" if build else "false"))
    c = [r for r in rows if r["counter"] == counter and pat.search(r["kernel"])]
    assert len(c) == 1, (counter, build, [r["kernel"] for r in c])
    return c[0]
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path))
for key, build in (("1000000x768_gaussian_sse_nq10000_k10_ef128", False), ("1000000x768_gaussian_sse_build_M16_efc200", True)):
    ent = tj[key]
    f, w = pick("FETCH_SIZE", build), pick("WRITE_SIZE", build)
    fk, wk = float(f["mean_value_KB"]), float(w["mean_value_KB"])
    per_launch = int((2 * fk + wk) * 1024)
    if build:
        ent["fetch_size_kb_mean_per_launch"], ent["write_size_kb_mean_per_launch"] = fk, wk
        ent["traffic_bytes_per_build"] = per_launch * int(f["dispatches"])
        ent["kernel"] = f'{f["kernel"]} (search half of the batch-synchronous build), all {f["dispatches"]} launches of one 1M x 768 build'
    else:
        ent["previous"] = f'{ent.get("traffic_bytes")} B at commit {ent.get("measured_at_commit")}'
        ent["fetch_size_kb"], ent["write_size_kb"], ent["traffic_bytes"] = fk, wk, per_launch
        ent["source"] = (f"profiles/{tag}_pmc_summary.csv (round 4, scripts/prof_bench.sh: separate --pmc passes; FETCH_SIZE doubled); kernel avg "
                         f"{float(f['mean_duration_ms']):.2f} ms under the counter pass, kernel stats in profiles/{tag}_kernel_stats.csv")
    ent["kernel_sources_sha256"] = kernel_sources_sha(ent["kernel_sources"])
    ent["measured_in_round"], ent["measured_at_commit"] = 4, commit
    print(key, per_launch, ent["kernel_sources_sha256"][:12])
json.dump(tj, open(tj_path, "w"), indent=1)
