"""How well the pipelined multi-exponentiations keep the accumulation kernel running (rocprofv3 --kernel-trace csv of bench.py):
the accumulation launches' durations, the idle gaps between consecutive ones, and what else ran in those gaps."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def name(r):
    m = re.search(r"(k_\w+)", r["Kernel_Name"]); return m.group(1) if m else r["Kernel_Name"][:30]
acc = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if name(r) == "k_accum28"])
acc = acc[len(acc) // 3:]                 # skip warm-up
durs = [(e - s) / 1e3 for s, e in acc]
gaps = [(acc[i + 1][0] - acc[i][1]) / 1e3 for i in range(len(acc) - 1)]
per = [(acc[i + 1][0] - acc[i][0]) / 1e3 for i in range(len(acc) - 1)]
print("accumulation launches:", len(acc), " mean duration %.1f us" % (sum(durs) / len(durs)), " mean start-to-start %.1f us" % (sum(per) / len(per)),
      " mean gap (next start - this end) %.1f us" % (sum(gaps) / len(gaps)))
lo, hi = acc[len(acc) // 2]
nxt = acc[len(acc) // 2 + 1]
print("one period in detail (t = 0 at an accumulation's start):")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > lo and s < nxt[1] and (e - s) > 3000:
        print("  %9.1f .. %9.1f us  %-24s %8.1f us  queue %s" % ((s - lo) / 1e3, (e - lo) / 1e3, name(r), (e - s) / 1e3, r.get("Queue_Id", "?")))
