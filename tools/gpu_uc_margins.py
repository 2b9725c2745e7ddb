"""Largest GPU-vs-oracle differences per use case (ORIG tree), to set test tolerances with margin."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from oracle import Oracle
from eepacc_mpc_casadi_matlab_amd.settings import Settings, SetVehicleParameters, default_opt
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
rec = load_golden("argonne_61505019_lead")
for case in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12):
    o = default_opt(); o["useCaseNum"] = case
    if case in (8, 9): o["argonne_lead"] = (rec["t"], rec["v_mph"])
    OPT = Settings(o, tree="ORIG", N_hor=20); V = SetVehicleParameters("ORIG")
    n = int(round(OPT["t_sim"] / OPT["Tvec"][0])) + 1
    if case in (8, 9, 10): s_tv, v_tv = np.asarray(OPT["s_tv"], float), np.asarray(OPT["v_tv"], float)
    else: s_tv, v_tv = np.full(n, np.inf), np.zeros(n)
    eng = Engine(OPT, V, device=0, max_batch=2)
    traj, status = eng.run_abmpc([OPT["s_init"]], [OPT["v_init"]], [OPT["a_minus1"]], s_tv[:, None], v_tv[:, None])
    tr = traj.cpu().numpy()[:, :, 0]
    ref, rst, _ = Oracle(OPT, V).run("ab", n, OPT["s_init"], OPT["v_init"], OPT["a_minus1"], s_tv.copy(), v_tv.copy())
    print(case, " ".join(f"{nm}:{np.abs(tr[:, OUT[nm]] - ref[:, OUT[nm]]).max():.1e}" for nm in ("s", "v", "a", "xi_v", "xi_f", "Fm", "Fb")), flush=True)
