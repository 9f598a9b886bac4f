"""print the headline fields of bench.py JSON lines:  python tools/show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e); continue
    rl = d.get("roofline") or {}
    rv = d.get("roofline_valu") or {}
    print("%-40s value=%.1f M/s  ms/step=%.3f  latency1=%.3f  accum_excl=%.3f  accum_pipe=%.3f  valu=%.2f  verified=%s" % (
        f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"], d.get("latency_ms_one_in_flight", float("nan")), rl.get("avg_launch_ms", float("nan")),
        rl.get("avg_launch_ms_pipelined", float("nan")), rv.get("frac", float("nan")), d.get("verified_bit_exact")))
