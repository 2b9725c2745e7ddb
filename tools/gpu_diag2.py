import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd._abi import OUT
N = 30; B = 4096; K = 6
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
st = status.cpu().numpy(); tr = traj.cpu().numpy()
ks, bs = np.nonzero(st)
print("bad:", list(zip(ks.tolist(), bs.tolist()))[:20])
# replay bad steps with the per-step operator (cold start) to read the raw solver status
Ts = 0.5
for k, b in list(zip(ks.tolist(), bs.tolist()))[:10]:
    s, v = tr[k, 0, b], tr[k, 1, b]
    a_prev = (v - tr[k - 1, 1, b]) / Ts if k > 0 else 0.0
    vtv = sc["v_tv"][k, b] if k > 0 else 0.0
    vtvp = sc["v_tv"][k - 1, b] if k > 1 else 0.0
    eng.reset()
    out, _, _, st1 = eng.ab_step([s], [v], [a_prev], [k * Ts], [sc["s_tv"][k, b]], [vtv], [(vtv - vtvp) / Ts if k > 0 else 0.0])
    it = eng.last_iterations(1)[0]
    print("k", k, "b", b, "cold status", st1.cpu().numpy()[0], "raw iters code", it, "inputs", s, v, a_prev, sc["s_tv"][k, b], vtv)
