"""Summarises the rocprofv3 --pmc passes of tools/gpu_sq.sh into one JSON: per kernel (the launches of the largest grid only: warm-ups and
self-checks are smaller) the average of every counter per launch, the kernel's average duration from the same passes' kernel trace, and
the derived figures the roofline discussion needs:
   quad-cycle counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*, SQ_BUSY_CYCLES: MI355X_MICROARCH.md) are reported raw and x4;
   clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration;   valu_insts_per_wave;   issue_cycles_per_valu = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU;
   wave_cycle split: ACTIVE_INST_ANY / WAIT_INST_ANY / WAIT_ANY as fractions of SQ_WAVE_CYCLES;
   occupancy = 4 * SQ_WAVE_CYCLES / (4 * SQ_BUSY_CYCLES ...) is left out: SQ_LEVEL_WAVES / SQ_BUSY_CU_CYCLES is reported as avg_waves_per_cu.
   python tools/sq_summary.py <pass dirs...> <out.json>"""
import collections, csv, glob, json, sys

dirs, out_path = sys.argv[1:-1], sys.argv[-1]
KEEP = ("k_accum28", "k_accum28_cxx", "k_ntt29_pass", "k_scatter_lds", "k_count_lds", "k_dimsum", "k_dimsum_mixed", "k_dimbits", "k_dimweight", "k_merge_a", "k_digits", "k_glv_digits", "k_glv_split", "k_accum_redo", "k_ms_pairs", "k_ms_hist", "k_ms_scatter", "k_ms_final")


def short(name):
    n = name.replace("vsp::(anonymous namespace)::", "").replace("vsp::", "").replace("void ", "")
    base = n.split("(")[0].split("<")[0].strip()
    g2 = "Affine28x2" in name or "Fp2x28" in name or "Fp2T" in name
    return base + (" [G2]" if g2 else "")


res = collections.defaultdict(lambda: {"counters": collections.defaultdict(list), "dur_ns": [], "grid": []})
for d in dirs:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        per_dispatch = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.split(" ")[0] not in KEEP:
                continue
            per_dispatch[(k, r["Dispatch_Id"])][r["Counter_Name"]] = per_dispatch[(k, r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            per_dispatch[(k, r["Dispatch_Id"])]["__grid"] = float(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        for (k, _), c in per_dispatch.items():
            res[k]["grid"].append(c.pop("__grid"))
            for name, val in c.items():
                res[k]["counters"][name].append(val)
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.split(" ")[0] in KEEP:
                res[k]["dur_ns"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), float(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))

out = {}
expanded = []
for k, v in sorted(res.items()):
    # launches of one kernel come in several sizes (self-checks, the passes of the staged sort, 2^20 and 2^22 problems): the two grid
    # sizes that took the most time in total, each as an entry of its own ("<kernel>" and "<kernel> #2")
    tot = collections.defaultdict(float)
    for d, g in v["dur_ns"]:
        tot[g] += d
    order = sorted(tot, key=lambda g: -tot[g])[:2] or [max(v["grid"]) if v["grid"] else 0]
    for rank, g in enumerate(order):
        expanded.append((k if rank == 0 else k + " #2", v, g))
for k, v, gmax in expanded:
    c = {}
    for name, vals in v["counters"].items():
        big = [x for x, g in zip(vals, v["grid"][:len(vals)]) if g == gmax] or vals
        c[name] = sum(big) / len(big)
    durs = [d for d, g in v["dur_ns"] if g == gmax] or [d for d, _ in v["dur_ns"]]
    dur = sum(durs) / len(durs) if durs else None
    e = {"launches_averaged": len([g for g in v["grid"] if g == gmax]) // max(1, len(dirs)), "grid_threads": gmax, "avg_duration_ms_under_pmc": dur / 1e6 if dur else None, "counters_per_launch": c}
    der = {}
    if dur and "GRBM_GUI_ACTIVE" in c:
        der["clock_ghz_from_GRBM_GUI_ACTIVE"] = c["GRBM_GUI_ACTIVE"] / 8.0 / dur
    if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
        der["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        der["issue_cycles_per_valu_inst"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
    if c.get("SQ_WAVE_CYCLES"):
        for name in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
            if name in c:
                der[name + "_over_WAVE_CYCLES"] = c[name] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_BUSY_CYCLES"):
            der["waves_resident_per_busy_SQ_cycle"] = c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if c.get("SQ_LEVEL_WAVES") and c.get("SQ_BUSY_CU_CYCLES"):
        der["avg_waves_per_cu_while_busy"] = c["SQ_LEVEL_WAVES"] / c["SQ_BUSY_CU_CYCLES"]
    if c.get("SQ_IFETCH") and c.get("SQ_INSTS_VALU"):
        der["ifetch_per_valu_inst"] = c["SQ_IFETCH"] / c["SQ_INSTS_VALU"]
    if dur and c.get("SQ_INSTS_VALU"):
        der["valu_wave_insts_per_s_per_simd"] = c["SQ_INSTS_VALU"] / (dur * 1e-9) / 1024.0
    e["derived"] = der
    out[k] = e
json.dump({"note": "rocprofv3 --pmc passes of tools/sq_workload.py (tools/gpu_sq.sh); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count "
                   "quad-cycles (MI355X_MICROARCH.md); counters are summed over the 8 XCDs; durations are under the profiler (kernels serialised)", "kernels": out},
          open(out_path, "w"), indent=1)
for k, e in out.items():
    print(k, "%.3f ms" % (e["avg_duration_ms_under_pmc"] or 0), {a: round(b, 3) for a, b in e["derived"].items()})
