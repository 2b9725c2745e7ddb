import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
log = open(os.path.join(ROOT, "gpurun_out", "small.log"), "w")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
OPT, V, s_tv, v_tv = make_case("ABO", 20)
G = load_golden("abo_abmpc")
eng = Engine(OPT, V, device=0, max_batch=64)
for B, n in ((1, 3), (5, 20), (5, 871), (9, 40)):
    stv = np.repeat(s_tv[:n, None], B, 1); vtv = np.repeat(v_tv[:n, None], B, 1)
    P("launch", B, n)
    traj, status = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
    torch.cuda.synchronize()
    tr = traj.cpu().numpy()
    P("done", B, n, "iters", eng.last_iterations(B), "max v err", np.abs(tr[:, OUT["v"], :] - G["v_opt"][:n, None]).max())
P("resume test")
B = 5
stv = np.repeat(s_tv[:871, None], B, 1); vtv = np.repeat(v_tv[:871, None], B, 1)
t1, _ = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv[:300], vtv[:300]); torch.cuda.synchronize(); P("part 1 ok")
t2, _ = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv[300:], vtv[300:], resume=True); torch.cuda.synchronize(); P("part 2 ok")
