#!/usr/bin/env python3
"""Times of the batched interior-point solver over the GPU operators (eepacc_mpc_casadi_matlab_amd.nlp.NlpSolver) on one
MI355X, in the two regimes where it converges:
  (a) 128 / 1024 routes of 60 s (120 intervals), different lead traces, from the car-following start;
  (b) 128 copies of the reference's 435 s route (870 intervals) from the saved IPOPT controls (local convergence).
One JSON line each: wall time of the batch, iterations, routes per second, the numpy oracle's time for ONE route."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import make_case, load_golden
    from oracle import nlp_oracle as M
    from eepacc_mpc_casadi_matlab_amd.nlp import NlpSolver, car_following_start
    OPT, V, s_tv, _ = make_case(tree="ABO")
    lm = V["lambda"] * V["m"]
    # (a)
    OPT["t_sim"] = 60.0
    P = M.NlpProblem(OPT, V, s_tv)
    sol = NlpSolver(OPT, V)
    t0 = time.perf_counter(); Ro = M.solve(P, M.NlpOptions(max_iter=100)); cpu_a = time.perf_counter() - t0
    for B in (128, 1024):
        rng = np.random.default_rng(1)
        offs = rng.uniform(0.0, 20.0, B)
        stv = np.stack([s_tv[:P.N] + o for o in offs])
        forces = np.stack([car_following_start(OPT, V, sol.tables, stv[i]) for i in range(B)])
        chi0 = np.tile(np.array([[0.0, 0.0, -P.drag(0.0, 0.0) / lm, 0.0]]), (B, 1))
        chi, u = sol.start_from_controls(stv, chi0, forces, margin=1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R = sol.solve(stv, chi, u, max_iter=100)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps({"case": "60 s routes from the car-following start", "routes": B, "intervals": P.N, "wall_s": dt,
                          "solved": int((R["status"] == 0).sum()), "iterations_max": int(R["iters"].max()),
                          "iterations_mean": float(R["iters"].double().mean()), "routes_per_s": B / dt,
                          "ms_per_iteration_of_the_batch": dt * 1e3 / max(1, int(R["iters"].max())),
                          "oracle_numpy_s_per_route": cpu_a, "oracle_iterations": Ro["iters"]}), flush=True)
    # (b)
    OPT, V, s_tv, _ = make_case(tree="ABO")
    P = M.NlpProblem(OPT, V, s_tv)
    G = load_golden("abo_nlp")
    sol = NlpSolver(OPT, V)
    B = 128
    forces = np.tile(np.stack([G["Fm_opt"], np.minimum(G["Fb_opt"], -1e-3)], axis=1)[None], (B, 1, 1))
    chi0 = np.tile(np.array([[0.0, 0.0, -P.drag(0.0, 0.0) / lm, 0.0]]), (B, 1))
    stv = np.tile(P.s_tv[None], (B, 1))
    chi, u = sol.start_from_controls(stv, chi0, forces, margin=1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = sol.solve(stv, chi, u, max_iter=60, mu_init=1e-4)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    U = np.stack([G[k] for k in ("Fm_opt", "Fb_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt")], axis=1)
    J_saved = P.eval_reference_form(G["s_opt"], G["v_opt"], G["theta_opt"], G["j_opt"], U)["J"]
    print(json.dumps({"case": "reference route (870 intervals) from the saved IPOPT controls", "routes": B, "wall_s": dt,
                      "solved": int((R["status"] == 0).sum()), "iterations_max": int(R["iters"].max()),
                      "ms_per_iteration_of_the_batch": dt * 1e3 / max(1, int(R["iters"].max())),
                      "J_rel_to_saved": float(R["J"][0]) / J_saved - 1.0, "reference_ipopt_tSolve_s_one_route": float(G["tSolve"][0])}), flush=True)


if __name__ == "__main__":
    main()
