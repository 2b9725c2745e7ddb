"""Baseline controller on the reference's use cases: where kernel and oracle part (debug aid)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import Oracle
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from eepacc_mpc_casadi_matlab_amd.settings import Settings, SetVehicleParameters, default_opt, Settings_BL
for case in [int(x) for x in sys.argv[1].split(",")]:
    o = default_opt(); o["useCaseNum"] = case
    OPT = Settings(o, tree="ABO", N_hor=20); V = SetVehicleParameters("ABO"); BL = Settings_BL(OPT)
    Ts = BL["Tvec"][0]; n = min(int(round(OPT["t_sim"] / Ts)) + 1, 400)
    if case == 10:
        s_tv, v_tv = np.asarray(OPT["s_tv"], dtype=np.float64)[:n], np.asarray(OPT["v_tv"], dtype=np.float64)[:n]
    else:
        s_tv, v_tv = np.full(n, np.inf), np.zeros(n)
    orc = Oracle(BL, V)
    ref, rst, _ = orc.run("ab", n, OPT["s_init"], OPT["v_init"], OPT["a_minus1"], s_tv.copy(), v_tv.copy())
    eng = Engine(BL, V, device=0, max_batch=n)
    traj, status = eng.run_abmpc(np.full(1, OPT["s_init"]), np.full(1, OPT["v_init"]), np.full(1, OPT["a_minus1"]), s_tv[:, None].copy(), v_tv[:, None].copy())
    tr = traj.cpu().numpy()[:, :, 0]; st = status.cpu().numpy()[:, 0]
    print("case", case, "steps", n, "gpu bad", np.where(st != 0)[0][:10], "orc bad", np.where(rst != 0)[0][:10])
    d = np.abs(tr[:, OUT["a_qp"]] - ref[:, OUT["a_qp"]])
    big = np.where(d > 1e-4)[0]
    print("  first steps with |d a_qp| > 1e-4:", big[:6])
    for k in big[:3]:
        print("   k", k, "s %.6f/%.6f v %.6f/%.6f a_qp %.6f/%.6f xi_f %.3e/%.3e cost %.6f/%.6f" % (tr[k, 0], ref[k, 0], tr[k, 1], ref[k, 1], tr[k, OUT["a_qp"]], ref[k, OUT["a_qp"]], tr[k, OUT["xi_f"]], ref[k, OUT["xi_f"]], tr[k, OUT["cost"]], ref[k, OUT["cost"]]))
        # same state through both per-step paths
        ap = (tr[k, 1] - tr[k - 1, 1]) / Ts if k else OPT["a_minus1"]
        r = orc.ab_step(s=float(tr[k, 0]), v=float(tr[k, 1]), a_prev=float(ap), t0=k * Ts, s_tv=float(s_tv[k]), v_tv=float(v_tv[k] if k else 0.0), a_tv_prev=float((v_tv[k] - (v_tv[k - 1] if k > 1 else 0.0)) / Ts if k else 0.0))
        e1 = Engine(BL, V, device=0, max_batch=1)
        out, _, _, s1 = e1.ab_step([tr[k, 0]], [tr[k, 1]], [ap], [k * Ts], [s_tv[k]], [v_tv[k] if k else 0.0], [(v_tv[k] - (v_tv[k - 1] if k > 1 else 0.0)) / Ts if k else 0.0], want_pred=False)
        oo = out.cpu().numpy()[:, 0]
        print("     at the kernel's state: oracle a_qp %.6f cost %.6f st %d | cold kernel a_qp %.6f cost %.6f st %d" % (r["out"][OUT["a_qp"]], r["out"][OUT["cost"]], r["status"], oo[OUT["a_qp"]], oo[OUT["cost"]], int(s1.cpu().numpy()[0])))
