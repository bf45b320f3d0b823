"""Condense the two `both` PMC passes of scripts/prof_n2v_halves.sh (FETCH_SIZE, WRITE_SIZE; tools/n2v_kernels.hip, 4 batches)
into profiles/<tag>_pmc_summary.csv and re-stamp profiles/traffic.json's Node2Vec entry with the sources measured.
usage: summarize_n2v_pmc.py <tag> <fetch counter csv> <write counter csv> [batches=4] [batches_per_run=640]"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402

tag, fpath, wpath = sys.argv[1:4]
batches = int(sys.argv[4]) if len(sys.argv) > 4 else 4
per_run = int(sys.argv[5]) if len(sys.argv) > 5 else 640


def load(path, cname):
    by = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] != cname or k.startswith("__amd_rocclr"):
            continue
        by.setdefault(k, []).append(float(r["Counter_Value"]))
    return by


F, W = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
rows, total = [], 0.0
for k in F:
    f, w = F[k], W.get(k, [])
    bytes_per_batch = (2.0 * sum(f) + sum(w)) * 1024.0 / batches  # FETCH_SIZE doubled on gfx950 (profiles/traffic.json _comment)
    total += bytes_per_batch
    rows.append((k[:100], len(f), sum(f) / len(f), sum(w) / max(1, len(w)), bytes_per_batch))
rows.sort(key=lambda r: -r[4])
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv"), "w") as out:
    out.write(f"kernel,dispatches_in_{batches}_batches,mean_FETCH_SIZE_KB,mean_WRITE_SIZE_KB,traffic_bytes_per_batch(2*FETCH+WRITE)\n")
    for r in rows:
        out.write('"%s",%d,%.3f,%.3f,%d\n' % r)
srcs = ["mn_n2v.hip", "mn_n2v_batched.hpp"]
tpath = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tpath))
ent = t["node2vec_er1M_20M_batched_default"]
ent.update({"traffic_bytes": int(total * per_run), "traffic_bytes_per_batch": int(total),
            "source": f"profiles/{tag}_pmc_summary.csv (round 3; tools/n2v_kernels.hip, separate --pmc passes; FETCH_SIZE doubled, "
                      "Infinity-Cache hits included)",
            "kernel_sources": srcs, "kernel_sources_sha256": kernel_sources_sha(srcs),
            "measured_at_commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()})
json.dump(t, open(tpath, "w"), indent=1)
print(f"{total / 1e9:.2f} GB per batch, {total * per_run / 1e12:.2f} TB per run; stamped {ent['kernel_sources_sha256'][:12]}")
