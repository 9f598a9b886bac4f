"""Kernel statistics out of a rocprofv3 rocpd SQLite database (default output format of rocprofv3 --kernel-trace):
   python tools/rocpd_stats.py <results.db> [csv_out]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                        f"from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for r in rows:
    lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for r in rows[:int(__import__("os").environ.get("TOP", "14"))]:
    name = r[0].replace("_ZN3vsp12_GLOBAL__N_1", "")[:60]
    print("%-60s n=%4d total=%9.1f us avg=%9.1f us %5.1f%%" % (name, r[1], r[2] / 1e3, r[3] / 1e3, 100.0 * r[2] / tot))
