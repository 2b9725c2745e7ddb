import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s1, make_s2
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from oracle import Oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
OPT, V, s_tv, v_tv = make_case("ABO", N)
G = load_golden("abo_abmpc")
B = 12
s1 = make_s1(B, G, s_tv, v_tv)
eng = Engine(OPT, V, device=0, max_batch=4096)
args = {k: s1[k] for k in ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")}
out, sp, vp, status = eng.ab_step(**args)
o = out.cpu().numpy(); st = status.cpu().numpy()
print("status", st, "iters", eng.last_iterations(B))
orc = Oracle(OPT, V)
t = time.time()
worst = {}
for i in range(B):
    r = orc.ab_step(**{k: float(v[i]) for k, v in args.items()})
    for n in ("Fm", "a", "xi_v", "xi_h", "xi_s", "xi_f", "cost"):
        worst[n] = max(worst.get(n, 0), abs(o[OUT[n], i] - r["out"][OUT[n]]))
print("oracle time per step", (time.time() - t) / B, "worst abs err", worst)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
Bb, K = 4096, 60
sc = make_s2(Bb, K + 10, lead["V_TO_2Hz"])
eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][:10], sc["v_tv"][:10]); torch.cuda.synchronize()
t = time.perf_counter()
traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][10:], sc["v_tv"][10:], resume=True); torch.cuda.synchronize()
dt = time.perf_counter() - t
print("N=%d closed loop B=%d K=%d: %.1f ms -> %.0f QP steps/s, bad %d" % (N, Bb, K, dt * 1e3, Bb * K / dt, int(status.sum().item())))
