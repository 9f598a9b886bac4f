"""Prints the kernel timeline of one 2^20 prove from a rocprofv3 --kernel-trace CSV of bench.py (diagnostic)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
nm = lambda r: r['Kernel_Name'].replace('vsp::(anonymous namespace)::', '').replace('vsp::', '')
mv = [i for i, r in enumerate(rows) if 'k_csr_matvec' in r['Kernel_Name']]
starts = mv[0::3]
sel, nxt = starts[4], starts[5]
first = min(int(r['Start_Timestamp']) for r in rows[sel - 12:sel + 1])
t0 = int(rows[sel]['Start_Timestamp'])
seg = [r for r in rows if t0 - 3_000_000 <= int(r['Start_Timestamp']) < int(rows[nxt]['Start_Timestamp']) - 3_000_000]
t0 = min(int(r['Start_Timestamp']) for r in seg)
print("span ms", (max(int(r['End_Timestamp']) for r in seg) - t0) / 1e6)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    agg[nm(r)[:46]][0] += 1; agg[nm(r)[:46]][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-48s n=%3d total %.3f ms" % (k, v[0], v[1]))
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d > 0.25 or 'ntt_pass' in r['Kernel_Name']:
        print("%8.3f -> %8.3f  q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6, r['Queue_Id'], nm(r)[:48]))
