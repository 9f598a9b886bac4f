"""Per-kernel durations of the LAST multi-exponentiation in a rocprofv3 --kernel-trace csv (diagnostic helper).
usage: kernel_times.py <kernel_trace.csv> [first-kernel-substring] [min-ms]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_ms_pairs"
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo = starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + 90]:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    n = (m.group(1) if m else r["Kernel_Name"])[:44]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= min_ms:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.2f} ms  {n:44s} grid {r['Grid_Size_X']:>10s} wg {r['Workgroup_Size_X']:>5s} vgpr {r['VGPR_Count']:>4s} lds {r['LDS_Block_Size']:>6s} {d:8.2f} ms")
