"""Phase timing of the dense QP kernel (needs a build with -DEEPACC_QP_TIMING)."""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine, load_library
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
log = open(os.path.join(ROOT, "gpurun_out", "qp_prof.log"), "w")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kind = sys.argv[4] if len(sys.argv) > 4 else "fb"
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, n, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
lib = load_library()
names = ["scan", "d=J'n", "z=J2d2", "r=R^-1d", "step", "add", "drop", "kkt", "chol+inv", "crash", "gi total"]
prof = (C.c_longlong * 16)()
stv = torch.as_tensor(sc["s_tv"], device="cuda"); vtv = torch.as_tensor(sc["v_tv"], device="cuda")
for k in range(n):
    t0 = time.time()
    eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[k:k + 1], vtv[k:k + 1], resume=(k > 0)); torch.cuda.synchronize()
    dt = time.time() - t0
    lib.eepacc_debug_qp_prof(prof, 1)
    it = eng.last_iterations(B)
    P(f"step {k}: {dt*1e3:.1f} ms, block 0 iterations {it[0]} (mean {it.mean():.0f})")
    P("   " + "  ".join(f"{nm} {prof[i] / 100.0 / 1e3:.2f}ms" for i, nm in enumerate(names)))
