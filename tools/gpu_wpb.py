import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
OPT, V, s_tv, v_tv = make_case("ABO", 20)
G = load_golden("abo_abmpc")
eng = Engine(OPT, V, device=0, max_batch=8)
B = 5; n = 60
stv = np.repeat(s_tv[:n, None], B, 1); vtv = np.repeat(v_tv[:n, None], B, 1)
traj, status = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
tr = traj.cpu().numpy()
print("WPB", os.environ.get("EEPACC_WPB"), "iters", eng.last_iterations(B), "status", status.cpu().numpy().sum())
e = np.abs(tr[:, OUT["v"], :] - G["v_opt"][:n, None])
print("first bad step per instance:", [int(np.argmax(e[:, i] > 1e-9)) if (e[:, i] > 1e-9).any() else -1 for i in range(B)])
print("xi_v err", np.abs(tr[:8, OUT["xi_v"], 0] - G["xi_v_opt"][:8]))
