"""One S2 instance, first n steps, through the structured FB kernel (debug builds print the solver's events)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
N = int(sys.argv[1]); i = int(sys.argv[2]); n = int(sys.argv[3])
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(i + 1, n, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=1)
idx = np.array([i])
traj, status = eng.run_fbmpc(sc["s0"][idx], sc["v0"][idx], sc["a_minus1"][idx], sc["s_tv"][:, idx].copy(), sc["v_tv"][:, idx].copy())
eng.synchronize()
print(traj.cpu().numpy()[:, :, 0], status.cpu().numpy()[:, 0])
