#!/usr/bin/env python3
"""Time of eepacc_nlp_riccati (the serial part of one interior-point iteration of RunOpt_NLP) on one MI355X:
the Newton system of the full 870-interval route at the car-following start, replicated for 128 / 1024 routes.
Prints one JSON line per batch size (ms per sweep, microseconds per stage of one route, routes per second)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import make_case
    from oracle import nlp_oracle as M
    from eepacc_mpc_casadi_matlab_amd.nlp import riccati_batched
    OPT, V, s_tv, _ = make_case(tree="ABO")
    P = M.NlpProblem(OPT, V, s_tv)
    chi, u = M.initial_point(P)
    r = M._stage_values(P, chi, u, 1e-5)[2]
    t = np.maximum(-r, 1e-2)
    Q, q, AB, c = M.assemble_newton(P, chi, u, 1.0 / t, t, np.zeros((P.N + 1, 4)), 1.0, 1e-5)
    t0 = time.perf_counter()
    for _ in range(3):
        M._riccati(Q, q, AB, c, 0.0)
    cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
    dev = torch.device("cuda", 0)
    for B in (128, 1024):
        Qd, qd, ABd, cd = (torch.from_numpy(np.ascontiguousarray(np.repeat(a[None], B, 0))).to(dev) for a in (Q, q, AB, c))
        reg = torch.zeros(B, dtype=torch.float64, device=dev)
        for _ in range(3):
            o = riccati_batched(Qd, qd, ABd, cd, reg, reg_scale=M.REG_SCALE)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o = riccati_batched(Qd, qd, ABd, cd, reg, reg_scale=M.REG_SCALE)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        bytes_sweep = B * P.N * (154 + 50 + 44 + 50 + 14) * 8
        print(json.dumps({"kernel": "k_riccati", "routes": B, "intervals": P.N, "ms_per_sweep": ms,
                          "us_per_stage_of_one_route": ms * 1e3 / (2 * P.N), "route_sweeps_per_s": B / (ms * 1e-3),
                          "hbm_GBps_algorithmic": bytes_sweep / (ms * 1e-3) / 1e9, "status_sum": int(o[3].sum().item()),
                          "cpu_numpy_ms_per_route_sweep": cpu_ms}), flush=True)


if __name__ == "__main__":
    main()
