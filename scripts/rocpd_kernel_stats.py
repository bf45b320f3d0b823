"""rocprofv3 (rocpd sqlite output) → per-kernel stats CSV like `--stats` prints.
usage: rocpd_kernel_stats.py <results.db> <out.csv>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
    for r in rows:
        f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
print("wrote", sys.argv[2], len(rows), "kernels")
