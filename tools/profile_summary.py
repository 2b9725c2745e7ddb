#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs that tools/run_profile.sh leaves under gpurun_out/prof_<tag>/ into profiles/:
    profiles/<name>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of the default bench command
    profiles/<name>_driver_kernel_stats.csv   same for the driver's `--steps 20 --warmup 5`
    profiles/<name>_summary.json              per-QP-step figures of the dominant kernel (what bench.py's roofline reads)

Usage: tools/profile_summary.py <prof dir> <name> <kernel substring> <batch> <steps per launch>
PMC values are summed over all dimensions of the kernel's LAST dispatch (the timed launch; the first is the warm-up).
FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them (MI355X_MICROARCH.md, HBM section: 8-byte-per-lane accesses
like this kernel's are uncalibrated, so the byte figure is indicative only)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof_dir, name, kern, batch, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
qp_steps = batch * steps
sys.path.insert(0, ROOT)
import bench as _bench                                             # noqa: E402  (source_hash of the kernel sources)
out = {"name": name, "kernel": kern, "qp_steps_per_launch": qp_steps, "source_hash": _bench.source_hash(),
       "build_flags": " ".join(os.environ.get("EEPACC_EXTRA_FLAGS", "").split()), "kernel_stats": [], "pmc": {}}


def one(pattern):
    f = sorted(glob.glob(os.path.join(prof_dir, pattern)))
    return f[0] if f else None


for tag, dst in (("stats", f"{name}_kernel_stats.csv"), ("stats_driver", f"{name}_driver_kernel_stats.csv")):
    f = one(f"{tag}_*kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(ROOT, "profiles", dst), "w") as g:
            w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows[:10])
        for r in rows:
            if kern in r["Name"]:
                out["kernel_stats"].append({"command": "default" if tag == "stats" else "--steps 20 --warmup 5",
                                            **{k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")}})
f = one("stats_*kernel_trace.csv")
if f:
    disp = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    out["dispatches"] = [dict(dur_ms=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                              vgpr=r.get("VGPR_Count"), agpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"),
                              lds_static=r.get("LDS_Block_Size"), scratch=r.get("Scratch_Size"),
                              grid=r.get("Grid_Size_X", r.get("Grid_Size")), wg=r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
                         for r in disp]
    if disp:
        out["timed_launch_ms"] = out["dispatches"][-1]["dur_ms"]
for f in sorted(glob.glob(os.path.join(prof_dir, "pmc*_counter_collection.csv"))):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
    if agg:
        last = max(k[0] for k in agg)
        for (d, cname), v in agg.items():
            if d == last:
                out["pmc"][cname] = v
p = out["pmc"]
if p:
    g = lambda k: p.get(k, 0.0)
    f64_wave_flops = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + 2.0 * g("SQ_INSTS_VALU_FMA_F64")
    # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = lanes active per VALU cycle, of 64
    lane_frac = min(1.0, g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))) if g("SQ_ACTIVE_INST_VALU") else None
    out["per_qp_step"] = {
        "valu_wave_insts": g("SQ_INSTS_VALU") / qp_steps, "salu_wave_insts": g("SQ_INSTS_SALU") / qp_steps,
        "lds_wave_insts": g("SQ_INSTS_LDS") / qp_steps,
        "fp64_add_wave_insts": g("SQ_INSTS_VALU_ADD_F64") / qp_steps, "fp64_mul_wave_insts": g("SQ_INSTS_VALU_MUL_F64") / qp_steps,
        "fp64_fma_wave_insts": g("SQ_INSTS_VALU_FMA_F64") / qp_steps, "fp64_trans_wave_insts": g("SQ_INSTS_VALU_TRANS_F64") / qp_steps,
        "fp64_flops_all_lanes": 64.0 * f64_wave_flops / qp_steps,
        "active_lane_fraction": lane_frac,
        "fp64_flops_active_lanes": (64.0 * f64_wave_flops / qp_steps) * (lane_frac if lane_frac else 1.0),
        "hbm_fetch_bytes": g("FETCH_SIZE") * 1024.0 / qp_steps, "hbm_write_bytes": g("WRITE_SIZE") * 1024.0 / qp_steps,
        "hbm_bytes_fetch_plus_write": (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0 / qp_steps,
        "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT") / qp_steps,
    }
    wc = g("SQ_WAVE_CYCLES")
    out["derived"] = {
        # SQ_* cycle counters count quad-cycles per wave (MI355X_MICROARCH.md): fractions of the waves' lifetime
        "valu_issue_util": g("SQ_ACTIVE_INST_VALU") / wc if wc else None,
        "wave_cycles_waiting": g("SQ_WAIT_ANY") / wc if wc else None,
        "wave_cycles_issue_stalled": g("SQ_WAIT_INST_ANY") / wc if wc else None,
        "wave_cycles_issuing_any": g("SQ_ACTIVE_INST_ANY") / wc if wc else None,
        "waves": g("SQ_WAVES"),
        "waves_per_simd": g("SQ_WAVES") / 1024.0,
        # share of the SIMD's time in which one of its resident waves has an instruction in flight
        "simd_issue_busy": (g("SQ_ACTIVE_INST_ANY") / wc) * (g("SQ_WAVES") / 1024.0) if wc else None,
    }
    if g("SQ_ACTIVE_INST_LDS") or g("SQ_INSTS_VMEM_RD"):
        # stall attribution.  ACTIVE_INST_* are per-wave quad-cycles with an instruction of that class in flight; the
        # LEVEL counters integrate the number of outstanding instructions, so LEVEL / INSTS is a mean latency (cycles)
        vm = g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")
        out["stall_split"] = {
            "active_valu": g("SQ_ACTIVE_INST_VALU") / wc if wc else None,
            "active_scalar": g("SQ_ACTIVE_INST_SCA") / wc if wc else None,
            "active_lds": g("SQ_ACTIVE_INST_LDS") / wc if wc else None,
            "active_vmem": g("SQ_ACTIVE_INST_VMEM") / wc if wc else None,
            "active_flat": g("SQ_ACTIVE_INST_FLAT") / wc if wc else None,
            "active_misc": g("SQ_ACTIVE_INST_MISC") / wc if wc else None,
            "wait_any": g("SQ_WAIT_ANY") / wc if wc else None,
            "wait_inst_any": g("SQ_WAIT_INST_ANY") / wc if wc else None,
            "wait_inst_lds": g("SQ_WAIT_INST_LDS") / wc if wc else None,
            "per_qp_step": {"vmem_rd_insts": g("SQ_INSTS_VMEM_RD") / qp_steps, "vmem_wr_insts": g("SQ_INSTS_VMEM_WR") / qp_steps,
                            "flat_insts": g("SQ_INSTS_FLAT") / qp_steps, "smem_insts": g("SQ_INSTS_SMEM") / qp_steps,
                            "branch_insts": g("SQ_INSTS_BRANCH") / qp_steps, "lds_insts": g("SQ_INSTS_LDS") / qp_steps},
            "mean_latency_cycles": {"vmem": g("SQ_INST_LEVEL_VMEM") / vm if vm else None,
                                    "lds": g("SQ_INST_LEVEL_LDS") / g("SQ_INSTS_LDS") if g("SQ_INSTS_LDS") else None,
                                    "smem": g("SQ_INST_LEVEL_SMEM") / g("SQ_INSTS_SMEM") if g("SQ_INSTS_SMEM") else None},
            # serial exposure: instructions x mean latency per QP step, as a share of a wave's lifetime per QP step
            "latency_cycles_per_qp_step": {"vmem": g("SQ_INST_LEVEL_VMEM") / qp_steps, "lds": g("SQ_INST_LEVEL_LDS") / qp_steps,
                                           "smem": g("SQ_INST_LEVEL_SMEM") / qp_steps},
            "wave_cycles_per_qp_step": 4.0 * wc / qp_steps,
        }
json.dump(out, open(os.path.join(ROOT, "profiles", f"{name}_summary.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k not in ("dispatches",)}, indent=1)[:4000])
