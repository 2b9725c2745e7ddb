#!/usr/bin/env python3
"""First GPU bring-up script: open-loop golden steps + closed loop vs golden (dev tool)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden, golden_step_inputs
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT

tree = sys.argv[1] if len(sys.argv) > 1 else "ABO"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
OPT, V, s_tv, v_tv = make_case(tree, N)
G = load_golden(f"{tree.lower()}_abmpc")
eng = Engine(OPT, V, device=0, max_batch=4096)
ks = np.arange(871)
inp = [golden_step_inputs(G, s_tv, v_tv, int(k)) for k in ks]
cols = {n: np.array([d[n] for d in inp]) for n in inp[0]}
t = time.time()
out, sp, vp, status = eng.ab_step(cols["s"], cols["v"], cols["a_prev"], cols["t0"], cols["s_tv"], cols["v_tv"], cols["a_tv_prev"])
torch.cuda.synchronize()
print("open-loop 871 cold steps: %.3f s" % (time.time() - t))
o = out.cpu().numpy(); st = status.cpu().numpy()
its = eng.last_iterations(871)
print("status nonzero:", int((st != 0).sum()), "iters mean/max", its.mean(), its.max())
if N == 20:
    for n in ("xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a", "DistHor"):
        g = G[n if n == "DistHor" else n + "_opt"]
        e = np.abs(o[OUT[n]] - g)
        print("  %-8s max err %.3e at k=%d" % (n, e.max(), e.argmax()))
# closed loop
B = 64
n_steps = 871
stv = np.repeat(s_tv[:n_steps, None], B, 1); vtv = np.repeat(v_tv[:n_steps, None], B, 1)
t = time.time()
traj, status = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
torch.cuda.synchronize()
dt = time.time() - t
print("closed loop B=%d x %d steps: %.3f s -> %.0f QP steps/s" % (B, n_steps, dt, B * n_steps / dt))
tr = traj.cpu().numpy(); st = status.cpu().numpy()
print("bad status:", int((st != 0).sum()), "total iters/instance:", eng.last_iterations(B)[:4])
if N == 20:
    for n in ("s", "v", "xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a", "DistHor"):
        g = G[n if n == "DistHor" else n + "_opt"]
        e = np.abs(tr[:, OUT[n], 0] - g)
        print("  %-8s max err %.3e at k=%d" % (n, e.max(), e.argmax()))
    print("  instance spread:", np.abs(tr - tr[:, :, :1]).max())
