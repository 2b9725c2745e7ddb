"""Event trace (library built with -DEEPACC_BL_TRACE) of the baseline controller's solve of use case 10, step 0."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from eepacc_mpc_casadi_matlab_amd.settings import Settings, Settings_BL, SetVehicleParameters, default_opt
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
o = default_opt(); o["useCaseNum"] = 10
OPT = Settings(o, tree="ABO", N_hor=20); V = SetVehicleParameters("ABO")
BL = Settings_BL(OPT); BL["bl_lp_eps"] = eps
eng = Engine(BL, V, device=0, max_batch=4)
s_tv = np.asarray(OPT["s_tv"], dtype=np.float64)[:1]; v_tv = np.zeros(1)
out, sp, vp, st = eng.ab_step(np.full(1, OPT["s_init"]), np.full(1, OPT["v_init"]), np.full(1, OPT["a_minus1"]), np.zeros(1),
                              s_tv, v_tv, np.zeros(1), want_pred=True)
torch.cuda.synchronize()
print("status", int(st[0]), "a_qp", float(out[OUT["a_qp"], 0]), "xi_f", float(out[OUT["xi_f"], 0]), "iters", eng.last_iterations(1))
print("s_tv", s_tv, "s0 v0", OPT["s_init"], OPT["v_init"])
print("v_pred", vp[:, 0].cpu().numpy())
