#!/usr/bin/env python3
"""Which cold starts reach a KKT point?  16 routes (lead speed trace scaled 0.9 ... 1.1, as bench.py --workload nlp draws them)
x a wide set of look-ahead / response-time starts, one batch, no early stop.  Prints the success matrix."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from conftest import make_case
    from eepacc_mpc_casadi_matlab_amd.nlp import NlpSolver, car_following_start
    from eepacc_mpc_casadi_matlab_amd.settings import Run_DrivingCycle
    OPT, V, _, _ = make_case("ABO")
    lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
    rng = np.random.default_rng(100)
    Rn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    facs = rng.uniform(0.9, 1.1, Rn)
    traces = np.stack([Run_DrivingCycle(OPT, V_TO_resampled=lead["V_TO_2Hz"] * f)[0] - OPT["TVlength"] for f in facs])
    sol = NlpSolver(OPT, V)
    N = sol.N
    restarts = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    starts = [(L, tc) for L in (40, 60, 90, 120, 160, 200, 300, 450) for tc in (2.0, 4.0, 8.0)]
    if len(sys.argv) > 3:
        starts = [(90, 4.0), (120, 2.0), (200, 2.0), (120, 4.0), (300, 2.0), (160, 8.0), (120, 8.0), (60, 2.0)]
    S = len(starts)
    stv = np.repeat(traces[:, :N], S, axis=0)
    forces = car_following_start(OPT, V, sol.tables, stv, lookahead=np.tile([L for L, _ in starts], Rn), tau=np.tile([t for _, t in starts], Rn))
    lm = V["lambda"] * V["m"]
    p0 = -(V["c_r"] * V["m"] * V["g"]) / lm
    chi, u = sol.start_from_controls(stv, np.tile(np.array([[0.0, 0.0, p0, 0.0]]), (Rn * S, 1)), forces, margin=1.0)
    t0 = time.perf_counter()
    R = sol.solve(stv, chi, u, max_iter=700, mu_init=1.0, restarts=restarts)
    torch.cuda.synchronize()
    st = R["status"].view(Rn, S).cpu().numpy()
    J = R["J"].view(Rn, S).cpu().numpy()
    it = R["iters"].view(Rn, S).cpu().numpy()
    ok = st == 0
    out = {"restarts": restarts, "wall_s": time.perf_counter() - t0, "routes": Rn, "starts": starts, "routes_solved": int(ok.any(axis=1).sum()),
           "per_start_successes": ok.sum(axis=0).tolist(), "per_route_successes": ok.sum(axis=1).tolist(),
           "best_J_spread_rel": [float((J[r][ok[r]].max() / J[r][ok[r]].min() - 1)) if ok[r].sum() > 1 else None for r in range(Rn)],
           "iters_of_successes_median": float(np.median(it[ok])) if ok.any() else None}
    # greedy cover: which few starts solve the most routes
    cover, left = [], ok.copy()
    for _ in range(8):
        gains = left.sum(axis=0)
        j = int(gains.argmax())
        if gains[j] == 0:
            break
        cover.append((starts[j], int(gains[j])))
        left[left[:, j]] = False
    out["greedy_cover"] = cover
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
