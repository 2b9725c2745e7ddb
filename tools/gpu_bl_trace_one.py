"""Event trace (library built with -DEEPACC_BL_TRACE) of ONE S2 instance's closed loop under the baseline controller (N = 30)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.settings import Settings_BL
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd.engine import Engine
inst = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(1, n, lead["V_TO_2Hz"], first_instance=inst)
OPT, V, _, _ = make_case("ABO", 30)
eng = Engine(Settings_BL(OPT), V, device=0, max_batch=1)
traj, st = eng.run_abmpc(*[torch.as_tensor(sc[k], device="cuda:0") for k in ("s0", "v0", "a_minus1", "s_tv", "v_tv")])
torch.cuda.synchronize()
print("iters", eng.last_iterations(1), "status", st.cpu().numpy().ravel())
