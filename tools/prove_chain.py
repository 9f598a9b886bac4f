"""The critical chain of ONE proof out of a rocprofv3 --kernel-trace CSV of bench.py: every kernel on the context's stream (witness_map
and the H multi-exponentiation), with its start, duration and the gap before it (diagnostic):
   python tools/prove_chain.py <..._kernel_trace.csv> [proof index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: r['Kernel_Name'].replace('vsp::(anonymous namespace)::', '').replace('vsp::', '').replace('void ', '')
mv = [i for i, r in enumerate(rows) if 'k_csr_matvec' in r['Kernel_Name']]
starts = mv[0::3]                                              # three mat-vecs per proof
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3               # the bench's single-context leg: proofs 1..5
a = int(rows[starts[k]]['Start_Timestamp']); b = int(rows[starts[k + 1]]['Start_Timestamp'])
q0 = rows[starts[k]]['Queue_Id']
chain = [r for r in rows if a - 200_000 <= int(r['Start_Timestamp']) < b - 200_000 and r['Queue_Id'] == q0]
print("proof %d: %.3f ms to the next proof's first mat-vec; %d kernels on the H chain's queue" % (k, (b - a) / 1e6, len(chain)))
prev = None; busy = 0.0
for r in chain:
    s = (int(r['Start_Timestamp']) - a) / 1e6; e = (int(r['End_Timestamp']) - a) / 1e6
    gap = s - prev if prev is not None else 0.0
    busy += e - s
    if e - s > 0.02 or gap > 0.02:
        print("%8.3f -> %8.3f  (%.3f ms, gap before %.3f)  %s" % (s, e, e - s, gap, nm(r)[:60]))
    prev = e
print("busy %.3f ms" % busy)
import collections
byq = collections.defaultdict(list)
for r in rows:
    if a - 200_000 <= int(r['Start_Timestamp']) < b - 200_000: byq[r['Queue_Id']].append(r)
for q, rs in sorted(byq.items()):
    big = max(rs, key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    print("queue %s: %3d kernels, first start %.3f, last end %.3f, longest %s (%.3f ms from %.3f)" % (q, len(rs), (int(rs[0]['Start_Timestamp']) - a) / 1e6,
          (max(int(r['End_Timestamp']) for r in rs) - a) / 1e6, nm(big)[:28], (int(big['End_Timestamp']) - int(big['Start_Timestamp'])) / 1e6, (int(big['Start_Timestamp']) - a) / 1e6))
