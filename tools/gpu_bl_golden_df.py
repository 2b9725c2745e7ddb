import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_case, load_golden, golden_step_inputs
from eepacc_mpc_casadi_matlab_amd.settings import Settings_BL
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
for tree in ("ABO", "ORIG"):
    OPT, V, s_tv, v_tv = make_case(tree, 20)
    G = load_golden(f"{tree.lower()}_blmpc")
    inps = [golden_step_inputs(G, s_tv, v_tv, k) for k in range(871)]
    names = ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")
    c = {n: np.array([d[n] for d in inps]) for n in names}
    keep = np.array([k not in (41, 6, 7, 8) for k in range(871)])
    eng = Engine(Settings_BL(OPT), V, device=0, max_batch=1024)
    o = eng.ab_step(**c)[0].cpu().numpy()
    dF = np.abs(o[OUT["Fm"]] - G["Fm_opt"]) + np.abs(o[OUT["Fb"]] - G["Fb_opt"])
    print(tree, "dF max", dF[keep].max(), "n>1e-4", int((dF[keep] > 1e-4).sum()), "n>2e-4", int((dF[keep] > 2e-4).sum()))
