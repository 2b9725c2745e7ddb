#!/usr/bin/env python3
"""Experiment: multi-start of the batched GPU solver on the reference's full route.  Starts: (i) the saved FBMPC force
trajectory delayed by 0..3 samples, blended with the car-following start and scaled; (ii) the saved IPOPT forces plus smooth
perturbations of growing amplitude (size of the basin around the saved solution).  Prints, per start, the final objective
relative to the saved IPOPT objective."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import make_case, load_golden
    from oracle import nlp_oracle as M
    from eepacc_mpc_casadi_matlab_amd.nlp import NlpSolver, car_following_start
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    OPT, V, s_tv, _ = make_case(tree="ABO")
    P = M.NlpProblem(OPT, V, s_tv)
    G, GF = load_golden("abo_nlp"), load_golden("abo_fbmpc")
    U = np.stack([G[k] for k in ("Fm_opt", "Fb_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt")], axis=1)
    J_saved = P.eval_reference_form(G["s_opt"], G["v_opt"], G["theta_opt"], G["j_opt"], U)["J"]
    sol = NlpSolver(OPT, V)
    N = P.N
    F_cf = car_following_start(OPT, V, sol.tables, P.s_tv).sum(axis=1)
    F_fb = (GF["Fm_opt"][:N] + np.minimum(GF["Fb_opt"][:N], 0.0))
    F_nlp = G["Fm_opt"] + np.minimum(G["Fb_opt"], 0.0)
    starts, labels = [], []
    for d in (0, 1, 2, 3):
        Fd = np.concatenate([np.repeat(F_fb[:1], d), F_fb[:N - d]])
        for beta in (0.0, 0.25, 0.5):
            for sc in (0.97, 1.0, 1.03):
                starts.append((1 - beta) * Fd * sc + beta * F_cf)
                labels.append("fb d=%d beta=%.2f scale=%.2f" % (d, beta, sc))
    rng = np.random.default_rng(0)
    k = np.arange(N)
    for amp in (5.0, 20.0, 50.0, 100.0, 200.0, 400.0, 800.0):
        for rep in range(2):
            noise = sum(rng.normal() * np.sin(2 * np.pi * k / per + rng.uniform(0, 6.28)) for per in (40.0, 90.0, 200.0, 400.0)) / 2.0
            starts.append(F_nlp + amp * noise)
            labels.append("ipopt + %g N" % amp)
    F = np.stack(starts)
    B = F.shape[0]
    Fm_min = -V["phi"] * V["T_m_max"] / V["eta_TF"] * 0.9
    forces = np.stack([np.maximum(F, Fm_min) + 1.0, np.minimum(F - np.maximum(F, Fm_min), 0.0) - 1e-3], axis=2)
    chi0 = np.tile(np.array([[0.0, 0.0, -P.drag(0.0, 0.0) / (V["lambda"] * V["m"]), 0.0]]), (B, 1))
    stv = np.tile(P.s_tv[None], (B, 1))
    chi, u = sol.start_from_controls(stv, chi0, forces, margin=0.5)
    t0 = time.perf_counter()
    R = sol.solve(stv, chi, u, max_iter=iters, mu_init=0.1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rel = [float(x) / J_saved - 1 for x in R["J"]]
    order = np.argsort(rel)
    out = {"starts": B, "iterations_budget": iters, "wall_s": dt, "best": [(labels[i], rel[i], int(R["status"][i]), int(R["iters"][i])) for i in order[:8]],
           "all": [(labels[i], rel[i], int(R["status"][i]), int(R["iters"][i])) for i in range(B)]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
