#!/usr/bin/env python3
"""Condense gpurun_out/prof_nlp/ (tools/run_profile_nlp.sh) into profiles/r02_nlp_eval_{kernel_stats.csv,summary.json}.
Per-unit figures: one unit = one (route, interval) = one thread of k_nlp_eval; PMC values are those of the kernel's
last dispatch, summed over its dimensions."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", "prof_nlp")
B, N, kern = 1024, 870, "k_nlp_eval"
units = B * N
out = {"name": "r02_nlp_eval", "kernel": kern, "units_per_launch": units, "pmc": {}}
f = glob.glob(os.path.join(D, "stats_*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
with open(os.path.join(ROOT, "profiles", "r02_nlp_eval_kernel_stats.csv"), "w") as g:
    w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows[:10])
for r in rows:
    if kern in r["Name"]:
        out["kernel_stats"] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")}
f = glob.glob(os.path.join(D, "stats_*kernel_trace.csv"))[0]
disp = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
if disp:
    r = disp[-1]
    out["dispatch"] = dict(vgpr=r.get("VGPR_Count"), agpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"),
                           scratch=r.get("Scratch_Size"), lds=r.get("LDS_Block_Size"),
                           grid=r.get("Grid_Size_X", r.get("Grid_Size")), wg=r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
for f in sorted(glob.glob(os.path.join(D, "pmc*_counter_collection.csv"))):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
    if agg:
        last = max(k[0] for k in agg)
        out["pmc"].update({c: v for (d, c), v in agg.items() if d == last})
p = out["pmc"]
avg_ns = float(out["kernel_stats"]["AverageNs"])
if "SQ_INSTS_VALU_FMA_F64" in p:
    act = p.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * p["SQ_ACTIVE_INST_VALU"]) if p.get("SQ_ACTIVE_INST_VALU") else 1.0
    wave_flop = p["SQ_INSTS_VALU_ADD_F64"] + p["SQ_INSTS_VALU_MUL_F64"] + 2 * p["SQ_INSTS_VALU_FMA_F64"]
    out["active_lane_fraction"] = act
    out["fp64_flop_per_unit"] = wave_flop * 64 * act / units
    out["fp64_flop_per_unit_all_lanes"] = wave_flop * 64 / units
    out["valu_insts_per_wave"] = p["SQ_INSTS_VALU"] / p["SQ_WAVES"] if p.get("SQ_WAVES") else None
    out["achieved_TFLOPs"] = out["fp64_flop_per_unit"] * units / avg_ns / 1e3
    out["frac_of_78.6"] = out["achieved_TFLOPs"] / 78.6
if "FETCH_SIZE" in p:
    out["fetch_bytes_per_unit"] = p["FETCH_SIZE"] * 1024 / units
if "WRITE_SIZE" in p:
    out["write_bytes_per_unit"] = p["WRITE_SIZE"] * 1024 / units
out["algorithmic_bytes_per_unit"] = 472
out["hbm_GBps_algorithmic"] = units * 472 / avg_ns
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_nlp_eval_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
