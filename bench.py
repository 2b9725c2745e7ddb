#!/usr/bin/env python3
"""Headline benchmark: ABMPC N=30 dense-QP MPC steps per second at batch 4096 (BASELINE.json configs[1]);
`--workload fbmpc` runs BASELINE configs[2] (FBMPC N=30, batch 4096) under the same contract.

One "step" = one receding-horizon step (measure -> estimate -> condense -> QP -> extract/allocate -> plant) for all
4096 synthetic S2 scenarios of a GPU (eepacc_mpc_casadi_matlab_amd/scenarios.py).  The timed region runs K consecutive
closed-loop steps (after W warm-up steps of the same simulation) with all inputs resident in HBM;
value = instances * K / time over all ranks.

    python bench.py --gpus 1 --steps 200 --warmup 200
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The `roofline` block states the bound that binds these fused kernels (primary definition):
fp64 vector issue.  HBM does not bind (about 0.2 KB of compulsory traffic per QP step) and there is no dense
contraction for MFMA.  achieved = fp64 flops the kernel executes per QP step (PMC counters of the committed profile of
this same command, profiles/r02_*_summary.json: SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 x 64 lanes x active-lane fraction)
x QP steps per launch / mean launch time measured live with HIP events.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP64_VALU_TFLOPS = 78.6        # MI355X vector fp64 (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)


def cpu_baseline(OPT, V, sc, kind="ab", n_inst=8, n_steps=200):
    """The oracle (literal dense condensing + dense active set) on the host on a bounded sample of
    the same workload: (i) one thread, (ii) one instance per thread on every host core
    (the C oracle is re-entrant and ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import Oracle
    orc = Oracle(OPT, V)
    n_steps = min(n_steps, sc["s_tv"].shape[0])

    def one(i):
        orc.run(kind, n_steps, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:n_steps, i].copy(), sc["v_tv"][:n_steps, i].copy())
        return n_steps

    t0 = time.perf_counter()
    done = sum(one(i) for i in range(n_inst))
    dt = time.perf_counter() - t0
    cores, cores_source = host_cores()
    n_all = min(4 * cores, sc["v0"].shape[0])
    t1 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        done_all = sum(ex.map(one, range(n_all)))
    dt_all = time.perf_counter() - t1
    return dict(value=done / dt, unit="QP steps/s", cores=1, kind="port",
                sample="%d S2 instances x %d closed-loop %sMPC steps, N=%d (oracle: literal dense condensing + dense dual active set, gcc -O3, 1 thread)"
                       % (n_inst, n_steps, "BL" if OPT.get("bl_mode") else kind.upper(), OPT["N_hor"]),
                all_cores={"value": done_all / dt_all, "cores": cores, "cores_source": cores_source,
                           "sample": "%d instances x %d steps, one instance per thread" % (n_all, n_steps)})


def host_cores():
    """Host cores this job may use: the affinity mask, capped by the cgroup CPU quota (a one-GPU box shows all 256 cores of
    the host in its mask but grants 16 of them)."""
    try:
        n = len(os.sched_getaffinity(0))
        src = "len(os.sched_getaffinity(0))"
    except AttributeError:
        n, src = (os.cpu_count() or 1), "os.cpu_count()"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                q = max(1, int(round(float(quota) / period)))
                if q < n:
                    n, src = q, "cgroup CPU quota (%s) below the affinity mask" % path
            break
        except Exception:
            continue
    return max(1, n), src


def source_hash():
    """Hash of the kernel sources the running library was built from (the profile tooling stores the same value)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "eepacc_mpc_casadi_matlab_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".inc", ".h", ".cpp")):
            h.update(fn.encode()); h.update(open(os.path.join(csrc, fn), "rb").read())
    return h.hexdigest()[:16]


def profile_figures(workload: str, N: int, driver_style: bool = False):
    """Per-QP-step figures of the dominant kernel from the committed PMC profile of this command: the newest round's
    summary for (workload, horizon[, the driver's short launch]).  Returns (summary | None, path, stale note | None):
    a summary taken on other kernel sources or build flags than the running library is still used (the flop count per
    QP step moves by a few per cent between kernel revisions) but flagged in the JSON line."""
    tag = {"abmpc": "ab", "fbmpc": "fb", "blmpc": "bl"}[workload]
    cands = []
    for rnd in ("r03", "r02"):
        if driver_style:
            cands.append("%s_%s_N%d_driver_summary.json" % (rnd, tag, N))
        cands.append("%s_%s_N%d_summary.json" % (rnd, tag, N))
    for name in cands:
        path = os.path.join(ROOT, "profiles", name)
        try:
            prof = json.load(open(path))
        except Exception:
            continue
        stale = None
        want = prof.get("source_hash")
        if want is None:
            stale = "profile predates source hashing (round 2 kernels)"
        elif want != source_hash():
            stale = "profile taken on kernel sources %s, running %s" % (want, source_hash())
        try:
            from eepacc_mpc_casadi_matlab_amd import engine
            flags = engine.load_library().eepacc_build_flags().decode()
            if prof.get("build_flags", "") != flags:
                stale = (stale + "; " if stale else "") + "build flags differ (%r vs %r)" % (prof.get("build_flags", ""), flags)
        except Exception:
            pass
        return prof, os.path.relpath(path, ROOT), stale
    return None, "profiles/r03_%s_N%d_summary.json" % (tag, N), None


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["abmpc", "fbmpc", "blmpc", "nlp"], default="abmpc",
                    help="abmpc: the headline (BASELINE.json configs[1]); fbmpc: configs[2]; blmpc: the baseline controller "
                         "(RunOpt_BLMPC, a handle with bl_mode = 1); same contract; nlp: BASELINE configs[4] -- RunOpt_NLP, "
                         "--batch routes per GPU (default there: 128 = 1024 / 8), one step = one cold-start solve of all of them")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--chunk", type=int, default=0, help="steps per kernel launch (0 = all K in one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (FBMPC, N=60, baseline controller, a few NLP routes) that the default "
                         "single-GPU headline command appends under \"secondary\"")
    return ap


def dist_setup(device=None, backend=None):
    """Rank, device and process group of this process -- shared by every workload.  One rank per GPU over RCCL ("nccl");
    EEPACC_DIST_BACKEND=gloo rehearses the multi-rank flow on fewer GPUs than ranks (ranks then share devices round-robin
    and the small result tensors are reduced on the host).  The process group is initialised before anything else
    touches the GPU.  Returns (rank, world, local_rank, backend, on_gpu, n_dev, dev, torch device)."""
    import torch
    import torch.distributed as dist
    from eepacc_mpc_casadi_matlab_amd.distributed import rank_world
    rank, world, local_rank = rank_world()
    backend = backend or os.environ.get("EEPACC_DIST_BACKEND", "nccl")
    on_gpu = device is None
    if on_gpu:
        n_dev = max(1, torch.cuda.device_count())
        dev = local_rank % n_dev
        d = torch.device("cuda", dev)
    else:
        n_dev, dev, d = 1, 0, device
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if on_gpu:
            torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=d)
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank, backend, on_gpu, n_dev, dev, d


def run_bench(args, make_engine=None, device=None, backend=None):
    """The benchmark job of one rank.  make_engine(OPT, V, device, B) builds the engine (default: the HIP engine);
    device / backend let the CPU test drive the same flow over gloo with a stand-in engine.  Returns the result
    dict on rank 0, None on the other ranks."""
    import torch
    import torch.distributed as dist
    from conftest import make_case
    from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
    from eepacc_mpc_casadi_matlab_amd._abi import OUT, OUT_N
    from eepacc_mpc_casadi_matlab_amd.distributed import shard_range, local_kpis, reduce_kpis, kpi_dict, max_over_ranks
    rank, world, local_rank, backend, on_gpu, n_dev, dev, d = dist_setup(device, backend)
    fb = args.workload == "fbmpc"
    bl = args.workload == "blmpc"
    K, W = args.steps, args.warmup
    N, B = args.horizon, args.batch
    OPT, V, _, _ = make_case("ABO", N)
    if bl:
        from eepacc_mpc_casadi_matlab_amd.settings import Settings_BL
        OPT = Settings_BL(OPT)
    Ts = float(OPT["Tvec"][0])
    lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
    lo, _ = shard_range(rank, world, B)                                  # rank r owns instances [rB, (r+1)B)
    sc = make_s2(B, W + K, lead["V_TO_2Hz"], first_instance=lo)
    if make_engine is None:
        from eepacc_mpc_casadi_matlab_amd.engine import Engine
        make_engine = lambda OPT_, V_, dev_, B_: Engine(OPT_, V_, device=dev_, max_batch=B_)
    eng = make_engine(OPT, V, dev, B)
    s_tv = torch.as_tensor(sc["s_tv"], device=d); v_tv = torch.as_tensor(sc["v_tv"], device=d)
    s0 = torch.as_tensor(sc["s0"], device=d); v0 = torch.as_tensor(sc["v0"], device=d); am1 = torch.as_tensor(sc["a_minus1"], device=d)
    chunk = args.chunk if args.chunk > 0 else K

    # outputs are preallocated once (the caller owns all buffers, include/eepacc.h); the timed window keeps its whole
    # trajectory for the key figures
    traj_all = torch.empty((max(K, W, 1), OUT_N, B), dtype=torch.float64, device=d)
    stat_all = torch.empty((max(K, W, 1), B), dtype=torch.int32, device=d)

    def run(lo_, hi_, resume, off):
        f = eng.run_fbmpc if fb else eng.run_abmpc
        n = hi_ - lo_
        return f(s0, v0, am1, s_tv[lo_:hi_], v_tv[lo_:hi_], resume=resume, out=(traj_all[off:off + n], stat_all[off:off + n]))

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    cutoff = float(OPT.get("cutOffDist", 1000.0))

    def kpis(n):
        # the key figures Main.m:203-263 prints, for this rank's instances over the window (energy: A10 post-processing)
        E = eng.postprocess(traj_all[:n])[3]
        return local_kpis(traj_all[:n], stat_all[:n], E, Ts, cutoff, OUT)

    # warm-up: W untimed steps of the simulation; also loads the code objects of every kernel the timed region
    # launches (ours and torch's small reductions, which are loaded lazily) and runs the collectives once
    if W > 0:
        run(0, W, False, 0)
        kw = kpis(W)
    else:
        traj_all[:1].zero_(); stat_all[:1].zero_()
        kw = kpis(1)
    kw = tuple(x if backend == "nccl" else x.cpu() for x in kw)
    reduce_kpis(*kw, world)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    if on_gpu:
        stream = torch.cuda.current_stream(d)      # the stream the engine launches on (engine._stream)
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if on_gpu:
        ev0.record(stream)
    launches = 0
    k = W
    while k < W + K:
        hi = min(k + chunk, W + K)
        run(k, hi, k > 0, k - W)
        launches += 1
        k = hi
    if on_gpu:
        ev1.record(stream)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0                  # exactly the K steps, bracketed by barrier + synchronize
    kernel_ms = ev0.elapsed_time(ev1) if on_gpu else dt * 1e3
    dt = max_over_ranks(dt, world, d if backend == "nccl" else None)
    # key-figure reduction of the job (the only collectives: SUM, MIN, MAX of three small tensors), after the clock
    # stopped: it is the job's final report, not part of the K steps
    t_k = time.perf_counter()
    kp = kpis(K)
    kp = tuple(x if backend == "nccl" else x.cpu() for x in kp)
    sums, mins, maxs = reduce_kpis(*kp, world)
    sync()
    kpi_ms = (time.perf_counter() - t_k) * 1e3
    if hasattr(eng, "synchronize"):
        eng.synchronize()                       # raises if the closed-loop kernel flagged a device-side failure
    iters = eng.last_iterations(B)
    neg_v = int((traj_all[:K, OUT["v"]] < -1e-11).sum().item())

    res = None
    if rank == 0:
        total_steps = world * B * K
        value = total_steps / dt
        nV, nC = (6 * N, 26 * N + 2) if fb else ((2 * N, 13 * N + 2) if bl else (5 * N, 14 * N + 2))
        bytes_mat = 8 * (nV * nV + nC * nV + 3 * nV + 2 * nC) + 8 * (nV + 1)     # SURVEY 8d, R-materialised
        bytes_fused = 152                                                          # SURVEY 8d, R-fused (compulsory)
        launch_s = (kernel_ms / 1e3) / launches
        qp_per_launch = B * (K / launches)
        qp_rate = qp_per_launch / launch_s                                         # QP steps/s of this rank's kernel
        driver_style = K / launches <= 40
        prof, prof_path, stale = profile_figures(args.workload, N, driver_style)
        kname = "k_fbs_run" if fb else "k_run_abmpc"
        if prof is not None:
            ps = prof["per_qp_step"]
            flops = ps["fp64_flops_active_lanes"]
            achieved = flops * qp_rate / 1e12
            dv = prof["derived"]
            roof = {"bound": "valu_fp64", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP64_VALU_TFLOPS,
                    "traffic": ps["hbm_bytes_fetch_plus_write"] * qp_per_launch,
                    "traffic_note": "FETCH_SIZE + WRITE_SIZE per QP step of the committed profile (%s; 8-byte-per-lane accesses, "
                                    "uncalibrated: MI355X_MICROARCH.md HBM section) x QP steps of this launch" % prof_path,
                    "definition": "SURVEY.md 8d primary: fp64 flops executed per QP step (%s: (ADD + MUL + 2 FMA)_F64 wave instructions "
                                  "x 64 lanes x active-lane fraction %.2f = %.0f flop) x QP steps per launch / mean launch time (HIP events, "
                                  "this run) against the vector-fp64 peak; HBM does not bind (compulsory traffic ~%d B per QP step) and "
                                  "the kernel has no dense contraction for MFMA" % (prof_path, ps["active_lane_fraction"], flops, bytes_fused),
                    "kernel": kname, "launches": launches, "launch_ms": launch_s * 1e3,
                    "valu_issue_util": dv["valu_issue_util"],
                    "fp64_pipe_util_all_lanes": ps["fp64_flops_all_lanes"] * qp_rate / 1e12 / PEAK_FP64_VALU_TFLOPS,
                    "wave_cycles_waiting": dv["wave_cycles_waiting"],
                    "occupancy_waves_per_simd": dv["waves_per_simd"],
                    # what actually limits the kernel: every resident wave is issuing an instruction
                    # for `wave_cycles_issuing_any` of its lifetime, so with w waves per SIMD the SIMD's issue port is taken
                    # for w x that share; only `fp64_share_of_valu` of the vector instructions are fp64 arithmetic
                    "simd_issue_busy": (dv.get("wave_cycles_issuing_any") or 0.0) * dv["waves_per_simd"],
                    "fp64_share_of_valu": (ps["fp64_add_wave_insts"] + ps["fp64_mul_wave_insts"] + ps["fp64_fma_wave_insts"]) / ps["valu_wave_insts"],
                    "profile": prof_path}
            if stale:
                roof["profile_stale"] = stale
        else:
            roof = {"bound": "valu_fp64", "achieved": None, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": None,
                    "traffic": None, "definition": "no committed PMC profile for this workload (%s)" % prof_path,
                    "kernel": kname, "launches": launches, "launch_ms": launch_s * 1e3}
        it_step = float(iters.mean()) / max(K / launches, 1)
        res = {
            "metric": "QP steps/sec (whole node), %s N=%d dense QP at batch %d" % ("FBMPC" if fb else ("BLMPC" if bl else "ABMPC"), N, B),
            "value": value, "unit": "QP steps/s", "n_gpus": min(world, n_dev) if on_gpu else 0, "ranks": world, "steps": K, "warmup": W,
            "ms_per_step": dt * 1e3 / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s N=%d fp64, batch=%d synthetic S2 ego/lead scenarios per GPU, closed loop"
                                   % ("FBMPC" if fb else ("BLMPC (baseline controller, LP)" if bl else "ABMPC"), N, B),
                       "batch_per_gpu": B, "horizon": N, "steps_per_launch": int(K / launches),
                       "parallelism": "instances sharded across %d GPU(s), no data-path collective" % world},
            "roofline": roof,
            # comparison figures only: what the reference's dense-QP API would move per QP step, and the
            # compulsory traffic of the fused step
            "comparison_figures": {
                "r_materialised_hbm": {"bytes_per_qp_step": bytes_mat, "as_if_GBps": bytes_mat * qp_rate / 1e9,
                                       "note": "bytes the reference's dense QP API moves; the fused kernels never materialise them"},
                "r_fused_hbm": {"bytes_per_qp_step": bytes_fused, "achieved_GBps": bytes_fused * qp_rate / 1e9}},
            "solver": {"mean_active_set_iterations_per_step": it_step, "bad_exits": int(round(float(sums[0]))),
                       "bad_exits_with_infeasible_measured_state": neg_v,
                       "note": "a measured speed below zero (after a stop behind a stopped lead) violates the hard row v_0 >= 0 "
                               "of the reference QP: that QP has no solution, the reference reports exitMessage = 1 as well"},
            "kpi": kpi_dict(sums, mins, maxs),
            "kpi_reduction_ms": kpi_ms,
        }
        if shared := (world > 1 and on_gpu and world > n_dev):
            res["note"] = "rehearsal: %d ranks share %d device(s)" % (world, n_dev)
        if not args.no_cpu_baseline and world == 1 and on_gpu:      # reported at N = 1 only
            res["cpu_baseline"] = cpu_baseline(OPT, V, sc, "fb", 2, 60) if fb else cpu_baseline(OPT, V, sc)      # ~10-25 s of host work
            res["cpu_baseline"]["host_cores_available"] = os.cpu_count()
    return res


def _compact(res):
    """The part of a result line a secondary entry keeps."""
    r = res["roofline"]
    keep = {k: res[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype")}
    keep["config"] = res["config"]["workload"]
    keep["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_ms",
                                              "simd_issue_busy", "occupancy_waves_per_simd", "profile", "profile_stale") if k in r}
    keep["solver"] = {k: res["solver"][k] for k in ("mean_active_set_iterations_per_step", "bad_exits", "bad_exits_with_infeasible_measured_state")}
    return keep


def secondary_measurements(args):
    """BASELINE configs 3-5 (and the baseline controller) measured in the same process AFTER the headline's clock stopped
    (default command, one rank): short launches of the driver's own shape so the whole command stays within minutes."""
    import copy
    out = {}
    plan = [("fbmpc_N30_b4096", dict(workload="fbmpc", horizon=30, batch=4096)),                # BASELINE configs[2]
            ("abmpc_N60_b8192", dict(workload="abmpc", horizon=60, batch=8192)),                # configs[3]: 65536 / 8 per GPU
            ("blmpc_N30_b4096", dict(workload="blmpc", horizon=30, batch=4096)),
            # the headline kernel when the launch is not bounded by the serial chains of its hardest instances (at batch 4096
            # the slowest instance's 20 steps take 7.1 of the launch's 7.3 ms): same command, four times
            # the instances
            ("abmpc_N30_b16384", dict(workload="abmpc", horizon=30, batch=16384)),
            ("fbmpc_N30_b16384", dict(workload="fbmpc", horizon=30, batch=16384)),
            ("abmpc_N60_b32768", dict(workload="abmpc", horizon=60, batch=32768))]
    for name, kw in plan:
        a = copy.copy(args)
        for k, v in kw.items():
            setattr(a, k, v)
        a.no_cpu_baseline = True
        a.chunk = 0
        t0 = time.perf_counter()
        try:
            out[name] = _compact(run_bench(a))
            out[name]["wall_s_incl_setup"] = time.perf_counter() - t0
        except Exception as e:                                    # a secondary entry never takes the headline down
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        a = copy.copy(args)
        a.workload, a.batch, a.steps, a.warmup = "nlp", NLP_SECONDARY_ROUTES, 1, 0
        a.nlp_max_iter = 1500              # bounds the wall time of this entry (the workload's own limit is NLPmaxIter = 5000)
        t0 = time.perf_counter()
        r = run_nlp_bench(a)
        out["nlp_%droutes" % NLP_SECONDARY_ROUTES] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "solver", "reference")}
        out["nlp_%droutes" % NLP_SECONDARY_ROUTES]["wall_s_incl_setup"] = time.perf_counter() - t0
    except Exception as e:
        out["nlp_%droutes" % NLP_SECONDARY_ROUTES] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


NLP_SECONDARY_ROUTES = 16           # BASELINE configs[4] asks 128 routes per GPU (profiles/r03_bench_nlp128.json: 128 of 128 in 125 s);
                                    # the default command carries 16 of them with the iteration limit 1500


NLP_STARTS = ((90, 4.0), (120, 2.0), (200, 2.0), (120, 4.0), (300, 2.0), (160, 8.0), (120, 8.0), (60, 2.0))


def nlp_traces(OPT, lead, rank, Rn):
    """Per-route lead vehicle of BASELINE configs[4]: the cycle's speeds scaled by U(0.9, 1.1) (seed 100 + rank)."""
    from eepacc_mpc_casadi_matlab_amd.settings import Run_DrivingCycle
    rng = np.random.default_rng(100 + rank)
    traces = []
    for f in rng.uniform(0.9, 1.1, Rn):
        s_tv, _ = Run_DrivingCycle(OPT, V_TO_resampled=lead["V_TO_2Hz"] * f)
        traces.append(s_tv - OPT["TVlength"])
    return np.stack(traces)


def run_nlp_bench(args, make_solver=None, device=None, backend=None):
    """BASELINE configs[4]: RunOpt_NLP for `--batch` routes per GPU (the reference's 435 s scenario with the lead vehicle's
    speed trace scaled per route), each solved from a cold start by the multi-start batch.  Routes shard across ranks with
    no data-path collective; one all-reduce (SUM) of (routes at a KKT point, sum of objectives, iterations) at the end.
    make_solver(OPT, V, dev) -> callable(traces) -> dict(status, J, iters) (torch tensors, one entry per route): default is
    the HIP solver; the CPU test drives the same flow over gloo with a stand-in."""
    import torch
    import torch.distributed as dist
    from conftest import make_case
    from eepacc_mpc_casadi_matlab_amd.distributed import max_over_ranks
    rank, world, local_rank, backend, on_gpu, n_dev, dev, d = dist_setup(device, backend)
    K = args.steps if args.steps != 200 else 1
    W = args.warmup if args.warmup != 200 else 0
    Rn = args.batch if args.batch != 4096 else 128
    OPT, V, _, _ = make_case("ABO")
    lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
    traces = nlp_traces(OPT, lead, rank, Rn)
    if make_solver is None:
        def make_solver(OPT_, V_, dev_):
            from eepacc_mpc_casadi_matlab_amd.nlp import NlpSolver, solve_routes
            sol = NlpSolver(OPT_, V_, device=dev_)
            mi = int(getattr(args, "nlp_max_iter", 0) or OPT_.get("NLPmaxIter", 5000))
            return lambda tr: solve_routes(sol, OPT_, V_, tr, NLP_STARTS, max_iter=mi)
    solve = make_solver(OPT, V, dev)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
    for _ in range(W):
        solve(traces)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        R = solve(traces)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = max_over_ranks(time.perf_counter() - t0, world, d if backend == "nccl" else None)
    kp = torch.stack([(R["status"] == 0).sum().double(), R["J"].double().sum(), R["iters"].double().sum()])
    kp = kp if (backend == "nccl" or not on_gpu) else kp.cpu()
    if world > 1:
        dist.all_reduce(kp, op=dist.ReduceOp.SUM)
    if rank != 0:
        return None
    total = world * Rn * K
    return {"metric": "RunOpt_NLP routes solved/s (whole node), 435 s route = 870 intervals, cold start", "value": total / dt,
            "unit": "routes/s", "n_gpus": min(world, n_dev) if on_gpu else 0, "ranks": world, "steps": K, "warmup": W, "ms_per_step": dt * 1e3 / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "RunOpt_NLP, %d routes per GPU x %d starts, 870 intervals, lead trace scaled per route" % (Rn, len(NLP_STARTS)),
                       "routes_per_gpu": Rn, "starts_per_route": len(NLP_STARTS)},
            "solver": {"routes_at_kkt_point": int(kp[0].item()), "routes": world * Rn, "mean_iterations": float(kp[2].item()) / (world * Rn),
                       "sum_objective": float(kp[1].item())},
            "reference": {"ipopt_tSolve_s_one_route": 190.4754463, "source": "ABO/savedNLPsol.mat (NLPsol.tSolve), other hardware"}}


def main():
    args = build_parser().parse_args()
    import torch.distributed as dist
    res = run_nlp_bench(args) if args.workload == "nlp" else run_bench(args)
    if (res is not None and args.workload == "abmpc" and not args.no_secondary and res.get("ranks") == 1 and res.get("n_gpus") == 1
            and args.horizon == 30 and args.batch == 4096):
        res["secondary"] = secondary_measurements(args)
    if res is not None:
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
