"""Working-set changes of cold-started QPs: the 871 saved ABMPC steps (ABO) as independent open-loop QPs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden, golden_step_inputs
from eepacc_mpc_casadi_matlab_amd.engine import Engine
OPT, V, s_tv, v_tv = make_case("ABO", 20)
G = load_golden("abo_abmpc")
inps = [golden_step_inputs(G, s_tv, v_tv, k) for k in range(871)]
c = {n: np.array([d[n] for d in inps]) for n in ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")}
eng = Engine(OPT, V, device=0, max_batch=1024)
eng.ab_step(**c); torch.cuda.synchronize()
eng.reset(); t0 = time.perf_counter(); out, _, _, st = eng.ab_step(**c); torch.cuda.synchronize(); dt = time.perf_counter() - t0
it = np.asarray(eng.last_iterations(871))
print(f"871 cold QPs (N = 20): {dt*1e3:.2f} ms; working-set changes mean {it.mean():.1f} median {np.median(it):.0f} max {it.max()}; failed {int((st.cpu().numpy() != 0).sum())}")
