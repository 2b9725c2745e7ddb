"""Baseline controller: kernel closed loop against the oracle's, first steps (debug aid)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_case, load_golden, golden_step_inputs
from oracle import Oracle
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from eepacc_mpc_casadi_matlab_amd.settings import Settings_BL
OPT, V, s_tv, v_tv = make_case("ABO", 20)
G = load_golden("abo_blmpc")
BL = Settings_BL(OPT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
eng = Engine(BL, V, device=0, max_batch=2)
traj, status = eng.run_abmpc(np.zeros(1), np.zeros(1), np.zeros(1), s_tv[:n, None].copy(), v_tv[:n, None].copy())
tr = traj.cpu().numpy()[:, :, 0]; st = status.cpu().numpy()[:, 0]
ref, rst, _ = Oracle(BL, V).run("ab", n, 0.0, 0.0, 0.0, s_tv[:n].copy(), v_tv[:n].copy())
for k in range(n):
    print(k, "v_gpu %.3e s_gpu %.3e v_orc %.3e" % (tr[k, OUT["v"]], tr[k, OUT["s"]], ref[k, OUT["v"]]), "gpu s %.6f v %.6f Fm %.4f a %.6f st %d | orc s %.6f v %.6f Fm %.4f a %.6f st %d | golden Fm %.4f" % (
        tr[k, OUT["s"]], tr[k, OUT["v"]], tr[k, OUT["Fm"]], tr[k, OUT["a"]], st[k],
        ref[k, OUT["s"]], ref[k, OUT["v"]], ref[k, OUT["Fm"]], ref[k, OUT["a"]], rst[k], G["Fm_opt"][k]))
# open loop on golden states with the per-step operator
ks = list(range(0, 60))
c = {nm: np.array([golden_step_inputs(G, s_tv, v_tv, k)[nm] for k in ks]) for nm in ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")}
out, sp, vp, status = eng2.ab_step(**c) if False else Engine(BL, V, device=0, max_batch=64).ab_step(**c)
o = out.cpu().numpy()
d = np.abs(o[OUT["Fm"]] - G["Fm_opt"][:60])
print("open loop |dFm| vs golden:", np.round(d, 6))
