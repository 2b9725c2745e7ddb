#!/usr/bin/env python3
"""Throughput of the RunOpt_NLP function evaluator (include/eepacc_nlp.h) on one MI355X: BASELINE config 5's
per-GPU share (1024 routes / 8 GPUs = 128) and the whole 1024 on one GPU, 870 intervals each (the reference route).
One "step" = one evaluation of objective, all rows, objective gradient and integrator Jacobian for the batch -- what
IPOPT calls back into once per iteration (RunOpt_NLP.m:505-510).  Prints one JSON line per batch size with the
roofline block (fp64 VALU: flops per (route, interval) from the PMC pass of profiles/r02_nlp_eval_summary.json when it
exists) and the numpy oracle timed on the host as cpu_baseline.

    python tools/gpu_nlp_bench.py [--batches 128,1024] [--steps 50] [--no-cpu]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="128,1024")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch
    from conftest import make_case, load_golden
    from eepacc_mpc_casadi_matlab_amd.nlp import NlpEvaluator
    OPT, V, s_tv, _ = make_case(tree="ABO")
    G = load_golden("abo_nlp")
    ev = NlpEvaluator(OPT, V)
    N = ev.N
    X0 = np.stack([G["s_opt"], G["v_opt"], G["theta_opt"], G["j_opt"]], axis=1)
    U0 = np.stack([G[k] for k in ("Fm_opt", "Fb_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt")], axis=1)
    prof = os.path.join(ROOT, "profiles", "r02_nlp_eval_summary.json")
    summ = json.load(open(prof)) if os.path.exists(prof) else None
    flop_unit = summ.get("fp64_flop_per_unit") if summ else None
    dev = torch.device("cuda", 0)
    for B in [int(b) for b in args.batches.split(",")]:
        rng = np.random.default_rng(5)
        X = np.repeat(X0[:, :, None], B, axis=2)
        U = np.repeat(U0[:, :, None], B, axis=2)
        X[1:, 1] = np.abs(X[1:, 1] + rng.normal(0, 0.3, (N, B)))
        U[:, 0] += rng.normal(0, 100.0, (N, B))
        stv = np.repeat(s_tv[:N, None], B, axis=1) + rng.normal(0, 1.0, (1, B))
        Xd, Ud, sd = (torch.from_numpy(a).to(dev) for a in (X, U, stv))
        out = ev.eval(sd, Xd, Ud)
        for _ in range(args.warmup):
            ev.eval(sd, Xd, Ud, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # evaluator runs on torch's stream
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            ev.eval(sd, Xd, Ud, out=out)
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms = e0.elapsed_time(e1) / args.steps
        units = N * B
        bytes_unit = (4 + 6 + 1) * 8 + (1 + 4 + ev.R + 10 + 6) * 8       # own node, controls, lead sample in; q, rows, gradients out
        line = {"metric": "route evaluations/s (RunOpt_NLP function evaluator, 870 intervals)", "value": B / (ms * 1e-3),
                "unit": "route evaluations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
                "wall_ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "RunOpt_NLP nlp_f/nlp_g/nlp_grad_f/integrator Jacobian, %d routes x %d intervals" % (B, N)},
                "units_per_s": units / (ms * 1e-3),
                "algorithmic_bytes_per_unit": bytes_unit, "hbm_GBps_algorithmic": units * bytes_unit / (ms * 1e-3) / 1e9}
        # which roofline binds: 472 algorithmic bytes against ~3.2 kflop fp64 per unit -> 6.7 flop/B; the machine balance
        # is 78.6 TF / 8 TB/s = 9.8 flop/B, so HBM binds (by a small margin); both fractions are reported
        traffic = None
        if summ:
            traffic = (summ.get("fetch_bytes_per_unit", 0) + summ.get("write_bytes_per_unit", 0)) * units
        line["roofline"] = {"bound": "hbm", "achieved": line["hbm_GBps_algorithmic"], "peak": 8000.0, "unit": "GB/s",
                            "frac": line["hbm_GBps_algorithmic"] / 8000.0, "traffic": traffic,
                            "definition": "algorithmic bytes per (route, interval) = 11 doubles in + 48 out = 472 B x units "
                            "/ mean time of one evaluation (both kernels; events on the launch stream); traffic = "
                            "(FETCH_SIZE + WRITE_SIZE) of the committed PMC passes (profiles/r02_nlp_eval_summary.json) per unit x units"}
        if flop_unit:
            ach = units * flop_unit / (ms * 1e-3) / 1e12
            line["valu_fp64"] = {"achieved_TFLOPs": ach, "peak": 78.6, "frac": ach / 78.6, "flop_per_unit": flop_unit}
        if not args.no_cpu and B == int(args.batches.split(",")[0]):
            from oracle import nlp_oracle as M
            P = M.NlpProblem(OPT, V, s_tv)
            n = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 10.0:
                i = n % B
                P.eval_reference_form(X[:, 0, i], X[:, 1, i], X[:, 2, i], X[:, 3, i], U[:, :, i])
                n += 1
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": n / dt, "unit": "route evaluations/s", "cores": 1, "kind": "port",
                                    "sample": "%d evaluations of one route (values only, numpy oracle, 1 thread)" % n}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
