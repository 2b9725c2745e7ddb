"""Timing of the FBMPC closed loop and its three kernels at BASELINE config 3 size."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
log = open(os.path.join(ROOT, "gpurun_out", "fb_time.log"), "w")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, n, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
stv = torch.as_tensor(sc["s_tv"], device="cuda"); vtv = torch.as_tensor(sc["v_tv"], device="cuda")
w = 4
eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[:w], vtv[:w]); torch.cuda.synchronize()
P("warm-up done; iterations/QP in last step", eng.last_iterations(B).mean())
t0 = time.time()
traj, status = eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[w:], vtv[w:], resume=True); torch.cuda.synchronize()
dt = time.time() - t0
P(f"FBMPC N={N} B={B}: {(n - w)} steps in {dt:.3f} s -> {B * (n - w) / dt:.0f} QP steps/s; bad exits {int((status != 0).sum())} of {status.numel()}")
P("iterations/QP in last step", eng.last_iterations(B).mean())
