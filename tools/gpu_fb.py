"""Dev check of the FBMPC closed loop on the GPU against the oracle closed loop."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from oracle.loader import Oracle
log = open(os.path.join(ROOT, "gpurun_out", "fb.log"), "w")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
tree = sys.argv[1] if len(sys.argv) > 1 else "ABO"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
OPT, V, s_tv, v_tv = make_case(tree, N)
orc = Oracle(OPT, V)
t0 = time.time()
ref, rst, rit = orc.run("fb", n, 0.0, 0.0, 0.0, s_tv[:n], v_tv[:n])
P("oracle", time.time() - t0, "s; bad", rst.sum())
eng = Engine(OPT, V, device=0, max_batch=8)
B = 3
stv = np.repeat(s_tv[:n, None], B, 1); vtv = np.repeat(v_tv[:n, None], B, 1)
t0 = time.time()
traj, status = eng.run_fbmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
torch.cuda.synchronize()
P("gpu", time.time() - t0, "s")
tr = traj.cpu().numpy(); st = status.cpu().numpy()
for name in ("s", "v", "Fm", "Fb", "a", "xi_v", "xi_h", "xi_s", "xi_f", "cost", "DistHor"):
    e = np.abs(tr[:, OUT[name], 0] - ref[:, OUT[name]])
    P(f"{name:8s} max err {e.max():.3e} at {e.argmax()}  (k>=1: {e[1:].max():.3e})  scale {np.abs(ref[:, OUT[name]]).max():.3e}")
e = np.abs(tr[:, OUT["Fm"], 0] + tr[:, OUT["Fb"], 0] - ref[:, OUT["Fm"]] - ref[:, OUT["Fb"]])
P("Fm+Fb max err", e.max())
P("status gpu", st[:, 0].tolist()); P("status orc", rst.tolist())
P("spread over batch", np.abs(tr[:, :, 0] - tr[:, :, 2]).max())
