"""Active-set iterations per MPC step of single S2 instances through the structured FB kernel (debug aid)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
N = 30
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
insts = [int(x) for x in sys.argv[1].split(",")]
nmax = int(sys.argv[2])
sc = make_s2(max(insts) + 1, nmax, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=1)
for i in insts:
    idx = np.array([i]); prev = 0; line = []
    for n in range(1, nmax + 1):
        t = time.time()
        traj, status = eng.run_fbmpc(sc["s0"][idx], sc["v0"][idx], sc["a_minus1"][idx], sc["s_tv"][:n, idx].copy(), sc["v_tv"][:n, idx].copy())
        eng.synchronize(); dt = time.time() - t
        tot = int(eng.last_iterations(1)[0]); st = int(status.cpu().numpy()[n - 1, 0])
        line.append(f"{tot - prev}{'*' if st else ''}")
        prev = tot
    print("inst", i, "iterations per step (* = failed):", " ".join(line), f"| last run of {nmax} steps {dt*1e3:.2f} ms", flush=True)
