"""ABMPC driver-style launch (5 warm-up steps, then 20 steps in one launch): per-instance totals of the launch.
Release library: working-set changes per instance; library built with -DEEPACC_DEBUG_TIMING: microseconds per instance."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
N = 30
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W, K = 5, 20
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, W + K, lead["V_TO_2Hz"], first_instance=0)
OPT, V, _, _ = make_case("ABO", N)
eng = Engine(OPT, V, device=0, max_batch=B)
d = "cuda:0"
s_tv = torch.as_tensor(sc["s_tv"], device=d); v_tv = torch.as_tensor(sc["v_tv"], device=d)
a3 = [torch.as_tensor(sc[k], device=d) for k in ("s0", "v0", "a_minus1")]
for rep in range(2):
    eng.run_abmpc(*a3, s_tv[:W], v_tv[:W]); torch.cuda.synchronize()
    t0 = time.perf_counter(); traj, st = eng.run_abmpc(*a3, s_tv[W:], v_tv[W:], resume=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
x = np.asarray(eng.last_iterations(B), dtype=np.float64)
print(f"B {B}: launch {dt*1e3:.2f} ms; per-instance total: mean {x.mean():.1f} median {np.median(x):.1f} p90 {np.percentile(x,90):.1f} p99 {np.percentile(x,99):.1f} max {x.max():.1f}  (sum/2048 waves {x.sum()/2048:.1f})")
top = np.argsort(-x)[:8]
print("largest:", [(int(i), float(x[i])) for i in top])
tr = traj.cpu().numpy()
for i in top[:3]:
    print(f" instance {i}: v {np.array2string(tr[:, OUT['v'], i], precision=2)}  a_qp {np.array2string(tr[:, OUT['a_qp'], i], precision=2)}  gap {np.array2string(sc['s_tv'][W:, i] - tr[:, OUT['s'], i], precision=1)}")
