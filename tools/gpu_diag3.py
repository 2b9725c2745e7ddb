import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden, golden_step_inputs
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
tree = sys.argv[1]; N = int(sys.argv[2])
OPT, V, s_tv, v_tv = make_case(tree, N)
G = load_golden(f"{tree.lower()}_abmpc")
eng = Engine(OPT, V, device=0, max_batch=1024)
inps = [golden_step_inputs(G, s_tv, v_tv, k) for k in range(871)]
c = {n: np.array([d[n] for d in inps]) for n in ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")}
out, sp, vp, status = eng.ab_step(**c)
st = status.cpu().numpy(); it = eng.last_iterations(871); o = out.cpu().numpy()
bad = np.nonzero(st)[0]
print("bad", bad.tolist())
for k in bad[:12]:
    code = int(it[k]); m = int(np.floor(code / 1e7 + 0.5)) if code >= 0 else -int(np.ceil(-code / 1e7))
    print(k, "raw", code, "Fm err", abs(o[OUT["Fm"], k] - G["Fm_opt"][k]))
e = np.abs(o[OUT["Fm"]] - G["Fm_opt"]); print("Fm max err", e.max(), e.argmax(), "iters(decoded mod 1e5) max", (it % 100000).max())
