#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace/stats and PMC passes) under gpurun_out/ into
profiles/<tag>_*.  Usage: tools/profile_summary.py <tag> <stats_dir> <pmc_dir>..."""
import csv, glob, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
PMC_KERNEL = os.environ.get("PMC_KERNEL", "k_run_abmpc")   # kernel whose counters are kept (last dispatch)
out = {"tag": tag, "kernels": [], "pmc": {}}
f = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    keep = [r for r in rows if "eepacc" in r["Name"] or "k_qp_dense" in r["Name"] or "k_fb_" in r["Name"]]
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows[:12])
    for r in keep:
        out["kernels"].append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")})
f = glob.glob(os.path.join(stats_dir, "**", "*kernel_trace.csv"), recursive=True)
if f:
    out["dispatches"] = [dict(kernel=r["Kernel_Name"][:60], dur_ms=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                              vgpr=r.get("VGPR_Count"), agpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"),
                              lds=r.get("LDS_Block_Size"), scratch=r.get("Scratch_Size"), grid=r.get("Grid_Size"), wg=r.get("Workgroup_Size"))
                         for r in csv.DictReader(open(f[0])) if "eepacc" in r["Kernel_Name"]][-12:]
for d in pmc_dirs:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if PMC_KERNEL in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
    last = max(k[0] for k in agg) if agg else None
    for (disp, name), v in agg.items():
        if disp == last:
            out["pmc"][name] = v
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
