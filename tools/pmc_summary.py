"""Summarises rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs, --output-format csv) of the same bench command
into profiles/<name>.json: per-kernel averages and the traffic of one k_accum launch = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 tallies
128-byte requests as 64 in FETCH_SIZE; /opt/skills/guides/MI355X_MICROARCH.md, HBM section).
   python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> "<command>" [key]     (key: k_accum_G1_2p20_plain | k_accum_G1_2p20_precomputed)
An existing <out.json> is extended, so the plain and the precomputed runs land in one file."""
import collections, csv, glob, json, os, sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("vsp::(anonymous namespace)::", "").replace("vsp::", "")[:80]
            agg[name].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "avg_KB": sum(v) / len(v), "max_KB": max(v)} for k, v in agg.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
ka = ([k for k in fetch if k.startswith("k_accum28")] + [k for k in fetch if k.startswith("void k_accum<Mont<FpP32>")])[0]
# the timed launches are the largest ones (warm-up and verification launches of other sizes are smaller)
fkb, wkb = fetch[ka]["max_KB"], write[ka]["max_KB"]
key = sys.argv[5] if len(sys.argv) > 5 else "k_accum_G1_2p20_plain"
alg = 128 * (1 << 20)
out = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
out[key] = {"command": sys.argv[4], "kernel": ka[:60], "FETCH_SIZE_KB_raw": fkb, "WRITE_SIZE_KB": wkb, "fetch_bytes_corrected": 2 * fkb * 1024, "write_bytes": wkb * 1024,
            "traffic_bytes_per_launch": 2 * fkb * 1024 + wkb * 1024, "algorithmic_bytes_per_launch": alg,
            "traffic_over_algorithmic": (2 * fkb * 1024 + wkb * 1024) / alg,
            "fetch_bytes_per_gather": 2 * fkb * 1024 / (16 * (1 << 20)),
            "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM) -> doubled; Infinity-Cache hits are "
                    "counted, so this is traffic past the XCD L2, not necessarily DRAM.  One 128-byte row of the 28-bit-limb table per gather "
                    "(16 gathers per base: one per window)",
            "per_kernel": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(key, ka[:40], "traffic per launch: %.3f GB = %.1f x algorithmic, %.0f B fetched per gather" % (out[key]["traffic_bytes_per_launch"] / 1e9, out[key]["traffic_over_algorithmic"], out[key]["fetch_bytes_per_gather"]))
