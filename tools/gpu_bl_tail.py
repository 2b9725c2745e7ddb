"""Baseline controller, driver-style launch: per-instance working-set changes (release library)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.settings import Settings_BL
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
N, B, W, K = 30, 4096, 5, 20
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, W + K, lead["V_TO_2Hz"], first_instance=0)
OPT, V, _, _ = make_case("ABO", N)
eng = Engine(Settings_BL(OPT), V, device=0, max_batch=B)
d = "cuda:0"
s_tv = torch.as_tensor(sc["s_tv"], device=d); v_tv = torch.as_tensor(sc["v_tv"], device=d)
a3 = [torch.as_tensor(sc[k], device=d) for k in ("s0", "v0", "a_minus1")]
for rep in range(2):
    eng.run_abmpc(*a3, s_tv[:W], v_tv[:W]); torch.cuda.synchronize()
    t0 = time.perf_counter(); traj, st = eng.run_abmpc(*a3, s_tv[W:], v_tv[W:], resume=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
x = np.asarray(eng.last_iterations(B), dtype=np.float64)
print(f"launch {dt*1e3:.1f} ms; per-instance changes: mean {x.mean():.0f} median {np.median(x):.0f} p90 {np.percentile(x,90):.0f} p99 {np.percentile(x,99):.0f} max {x.max():.0f} (sum/2048 {x.sum()/2048:.0f})")
top = np.argsort(-x)[:6]
print("largest:", [(int(i), float(x[i])) for i in top])
tr = traj.cpu().numpy()
for i in top[:2]:
    print(f" instance {i}: v {np.array2string(tr[:, OUT['v'], i], precision=2)} gap {np.array2string(sc['s_tv'][W:, i] - tr[:, OUT['s'], i], precision=1)}")
