"""Where do the ~10 ms between the HIP-event bracket and the kernel's own duration go?"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd._abi import OUT_N
N, B, W, K = 30, 4096, 20, 200
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, W + K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
d = torch.device("cuda", 0)
s_tv = torch.as_tensor(sc["s_tv"], device=d); v_tv = torch.as_tensor(sc["v_tv"], device=d)
s0 = torch.as_tensor(sc["s0"], device=d); v0 = torch.as_tensor(sc["v0"], device=d); am1 = torch.as_tensor(sc["a_minus1"], device=d)
buf = (torch.empty((K, OUT_N, B), dtype=torch.float64, device=d), torch.empty((K, B), dtype=torch.int32, device=d))
for rep in range(4):
    eng.run_abmpc(s0, v0, am1, s_tv[:W], v_tv[:W], out=buf); torch.cuda.synchronize()
    st = torch.cuda.current_stream(d)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t0 = time.perf_counter(); e[0].record(st)
    a = s_tv[W:W + K]; b = v_tv[W:W + K]
    t1 = time.perf_counter(); e[1].record(st)
    traj, status = eng.run_abmpc(s0, v0, am1, a, b, resume=True, out=buf)
    t2 = time.perf_counter(); e[2].record(st)
    x = status.sum()
    e[3].record(st)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"rep {rep}: cpu slice {1e3*(t1-t0):.3f} enqueue {1e3*(t2-t1):.3f} total {1e3*(t3-t0):.3f} ms | gpu e0-e1 {e[0].elapsed_time(e[1]):.3f} e1-e2 {e[1].elapsed_time(e[2]):.3f} e2-e3 {e[2].elapsed_time(e[3]):.3f}", flush=True)
