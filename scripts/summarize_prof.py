"""Condense rocprofv3 outputs under gpurun_out/ into small committed files under profiles/.
usage: summarize_prof.py <tag> <kernel_stats.csv> [<pmc_fetch counter csv> <pmc_write counter csv>]"""
import csv, sys, shutil, os, statistics as st
tag = sys.argv[1]
os.makedirs("profiles", exist_ok=True)
shutil.copy(sys.argv[2], f"profiles/{tag}_kernel_stats.csv")
if len(sys.argv) > 4:
    out = []
    for path, cname in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
        rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == cname]
        by = {}
        for r in rows:
            by.setdefault(r["Kernel_Name"], []).append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size"])))
        for k, v in by.items():
            if not k.startswith("void k_") and not k.startswith("k_"):
                continue
            out.append((cname, k, len(v), st.mean(x[0] for x in v), max(x[0] for x in v), st.mean(x[1] for x in v)))
    with open(f"profiles/{tag}_pmc_summary.csv", "w") as f:
        f.write("counter,kernel,dispatches,mean_value_KB,max_value_KB,mean_duration_ms\n")
        for o in out:
            f.write('%s,"%s",%d,%.3f,%.3f,%.4f\n' % o)
print("written profiles/%s_*" % tag)
