"""Which instances carry the tail of an FBMPC launch: iterations per instance over steps [W, W+K) (debug aid)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
N, B = 30, 4096
W, K = int(sys.argv[1]), int(sys.argv[2])
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, W + K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
stv = torch.as_tensor(sc["s_tv"], device="cuda"); vtv = torch.as_tensor(sc["v_tv"], device="cuda")
eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[:W], vtv[:W]); eng.synchronize()
t = time.time()
traj, status = eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[W:], vtv[W:], resume=True); eng.synchronize()
dt = time.time() - t
it = eng.last_iterations(B).astype(np.int64)
st = status.cpu().numpy()
print(f"steps {W}..{W+K-1}: {dt*1e3:.1f} ms, iterations per instance: mean {it.mean():.1f} median {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()}")
top = np.argsort(-it)[:12]
for i in top:
    print(" inst", int(i), "iterations", int(it[i]), "failed steps", int((st[:, i] != 0).sum()), "first failed", int(np.argmax(st[:, i] != 0)) if (st[:, i] != 0).any() else None, "v0", float(sc["v0"][i]))
print("TOP", ",".join(str(int(i)) for i in top[:6]))
