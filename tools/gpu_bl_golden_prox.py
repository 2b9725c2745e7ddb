"""Baseline controller: the 871 saved steps (ABO) as open-loop LPs for several (bl_lp_eps, bl_prox_iter); the worst steps
are replayed alone (a library built with -DEEPACC_BL_TRACE prints the proximal iterations)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden, golden_step_inputs
from eepacc_mpc_casadi_matlab_amd.settings import Settings_BL
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
OPT, V, s_tv, v_tv = make_case("ABO", 20)
G = load_golden("abo_blmpc")
inps = [golden_step_inputs(G, s_tv, v_tv, k) for k in range(871)]
names = ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")
c = {n: np.array([d[n] for d in inps]) for n in names}
keep = np.array([k not in (41, 6, 7, 8) for k in range(871)])
cost0 = None
for eps, prox in [(1e-4, -1), (1e-2, 40), (0.1, 40), (1.0, 40), (3.0, 40)]:
    BL = Settings_BL(OPT); BL["bl_lp_eps"] = eps; BL["bl_prox_iter"] = prox
    eng = Engine(BL, V, device=0, max_batch=1024)
    out, sp, vp, st = eng.ab_step(**c)
    o = out.cpu().numpy()
    dF = np.abs(o[OUT["Fm"]] - G["Fm_opt"]) + np.abs(o[OUT["Fb"]] - G["Fb_opt"])
    dF[~keep] = 0
    worst = np.argsort(-dF)[:5]
    if cost0 is None: cost0 = o[OUT['cost']].copy()
    print('   cost - cost(first config) at the worst steps:', (o[OUT['cost']] - cost0)[worst], ' cost:', o[OUT['cost']][worst], ' xi_f:', o[OUT['xi_f']][worst])
    print(f"eps {eps:g} prox {prox}: failed {int((st.cpu().numpy() != 0).sum())}  dF max {dF.max():.3e}  >2e-4: {int((dF > 2e-4).sum())}  >1e-2: {int((dF > 1e-2).sum())}  worst {list(worst)}  iters mean {np.mean(np.asarray(eng.last_iterations(871)) % 10000):.1f}  prox histogram {np.bincount(np.asarray(eng.last_iterations(871)) // 10000)}", flush=True)
    if len(sys.argv) > 1 and prox > 0:
        e1 = Engine(BL, V, device=0, max_batch=4)
        k = int(worst[0])
        o1 = e1.ab_step(**{n: c[n][k:k + 1] for n in names})[0]
        torch.cuda.synchronize()
        print(f"   replay step {k}: a_qp {float(o1[OUT['a_qp'], 0]):.9f}  Fm {float(o1[OUT['Fm'], 0]):.6f} (saved {G['Fm_opt'][k]:.6f})  Fb {float(o1[OUT['Fb'], 0]):.6f} (saved {G['Fb_opt'][k]:.6f})", flush=True)
