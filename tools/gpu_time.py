import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd._abi import OUT, OUT_N
N = 30; B = 4096; W = 10; K = 100
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, W + K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
d = torch.device("cuda", 0)
s_tv = torch.as_tensor(sc["s_tv"], device=d); v_tv = torch.as_tensor(sc["v_tv"], device=d)
s0 = torch.as_tensor(sc["s0"], device=d); v0 = torch.as_tensor(sc["v0"], device=d); am1 = torch.as_tensor(sc["a_minus1"], device=d)
buf = (torch.empty((K, OUT_N, B), dtype=torch.float64, device=d), torch.empty((K, B), dtype=torch.int32, device=d))
def sync(): torch.cuda.synchronize()
for rep in range(3):
    eng.run_abmpc(s0, v0, am1, s_tv[:W], v_tv[:W], out=buf); sync()
    t0 = time.perf_counter()
    traj, status = eng.run_abmpc(s0, v0, am1, s_tv[W:], v_tv[W:], resume=True, out=buf)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    kpi = torch.stack([status.sum().to(torch.float64), traj[-1, OUT["s"]].sum(), (traj[:, OUT["a"]] ** 2).sum()]); sync()
    t3 = time.perf_counter()
    print("rep", rep, "launch call %.1f ms, kernel wait %.1f ms, kpi %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
