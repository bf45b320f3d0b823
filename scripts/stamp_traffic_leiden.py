#!/usr/bin/env python3
"""Re-stamp the two synchronous-Leiden entries of profiles/traffic.json from the raw rocprofv3 counter files of scripts/prof_leiden.sh
(ONE run_leiden per pass): FETCH_SIZE (doubled: gfx950, MI355X_MICROARCH.md) + WRITE_SIZE summed over EVERY launch of the pass.
usage: stamp_traffic_leiden.py <unweighted tag> <weighted tag> <commit>   (tags as given to prof_leiden.sh: gpurun_out/prof_<tag>_{f,w})"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402


def total(tag, p, cname):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{p}", "*counter_collection.csv"))
    assert len(f) == 1, f
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == cname]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows)


tu, tw, commit = sys.argv[1:4]
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path))
for key, tag, name in (("leiden_lfr500k_unweighted_sync_default", tu, "unweighted"), ("leiden_lfr500k_weighted_sync_default", tw, "weighted")):
    ent = tj[key]
    (fk, nf), (wk, _) = total(tag, "f", "FETCH_SIZE"), total(tag, "w", "WRITE_SIZE")
    ent["previous"] = {"traffic_bytes": ent["traffic_bytes"], "note": f"commit {ent['measured_at_commit']} (the Leiden kernels are the same; mn_graph.hip "
                       "changed in its Brandes section, which made the file-level stamp stale, so the passes were run again)"}
    ent["fetch_size_kb"], ent["write_size_kb"], ent["traffic_bytes"] = fk, wk, int((2 * fk + wk) * 1024)
    ent["source"] = (f"profiles/r04_leiden_500k_9M_{name}_pmc_summary.csv (round 4, last session: scripts/prof_leiden.sh, separate --pmc passes, "
                     f"{nf} launches of one run_leiden; FETCH_SIZE doubled)")
    ent["kernel_sources_sha256"] = kernel_sources_sha(ent["kernel_sources"])
    ent["measured_in_round"], ent["measured_at_commit"] = 4, commit
    print(key, ent["traffic_bytes"], nf)
json.dump(tj, open(tj_path, "w"), indent=1)
