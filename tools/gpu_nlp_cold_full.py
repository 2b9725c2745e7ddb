#!/usr/bin/env python3
"""Experiment: the reference's full 435 s route from cold starts on the GPU solver with a long iteration budget
(car-following start and the saved FBMPC force trajectory).  Prints objective / status against the saved IPOPT objective."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import make_case, load_golden
    from oracle import nlp_oracle as M
    from eepacc_mpc_casadi_matlab_amd.nlp import NlpSolver, car_following_start
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    OPT, V, s_tv, _ = make_case(tree="ABO")
    P = M.NlpProblem(OPT, V, s_tv)
    G, GF = load_golden("abo_nlp"), load_golden("abo_fbmpc")
    U = np.stack([G[k] for k in ("Fm_opt", "Fb_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt")], axis=1)
    J_saved = P.eval_reference_form(G["s_opt"], G["v_opt"], G["theta_opt"], G["j_opt"], U)["J"]
    sol = NlpSolver(OPT, V)
    N = P.N
    f_cf = car_following_start(OPT, V, sol.tables, P.s_tv)
    f_fb = np.stack([GF["Fm_opt"][:N], np.minimum(GF["Fb_opt"][:N], -1e-3)], axis=1)
    # the FBMPC trajectory delayed by one step (the NLP reads the lead position one sample earlier, RunOpt_NLP.m:488-499)
    f_fbd = np.concatenate([f_fb[:1], f_fb[:-1]])
    forces = np.stack([f_cf, f_fb, f_fbd])
    B = forces.shape[0]
    chi0 = np.tile(np.array([[0.0, 0.0, -P.drag(0.0, 0.0) / (V["lambda"] * V["m"]), 0.0]]), (B, 1))
    stv = np.tile(P.s_tv[None], (B, 1))
    for margin, mu0 in ((1.0, 1.0), (0.1, 0.1)):
        chi, u = sol.start_from_controls(stv, chi0, forces, margin=margin)
        t0 = time.perf_counter()
        R = sol.solve(stv, chi, u, max_iter=iters, mu_init=mu0)
        torch.cuda.synchronize()
        print(json.dumps({"margin": margin, "mu_init": mu0, "iters": R["iters"].tolist(), "status": R["status"].tolist(),
                          "J_rel_to_saved": [float(x) / J_saved - 1 for x in R["J"]], "kkt": R["kkt"].tolist(),
                          "wall_s": time.perf_counter() - t0}), flush=True)


if __name__ == "__main__":
    main()
