"""Baseline controller (LP): sweep of the regularising curvature bl_lp_eps on the bench's S2 scenarios and on use case 10.
Prints failed steps, time of a 25-step launch and the difference of the applied acceleration to the default (1e-4)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.settings import Settings, Settings_BL, SetVehicleParameters, default_opt
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT

N, B, K = 30, 4096, 25
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, K, lead["V_TO_2Hz"], first_instance=0)
ref = None
for eps, prox in [(1e-4, -1), (0.01, 40), (0.1, 40), (1.0, 40)]:
    OPT, V, _, _ = make_case("ABO", N)
    BL = Settings_BL(OPT); BL["bl_lp_eps"] = eps; BL["bl_prox_iter"] = prox
    eng = Engine(BL, V, device=0, max_batch=B)
    args = [torch.as_tensor(sc[k], device="cuda:0") for k in ("s0", "v0", "a_minus1", "s_tv", "v_tv")]
    eng.run_abmpc(*args); torch.cuda.synchronize()
    t0 = time.perf_counter(); traj, st = eng.run_abmpc(*args); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    a = traj[:, OUT["a_qp"]].cpu().numpy(); stn = st.cpu().numpy()
    if ref is None: ref = a
    d = np.abs(a - ref); ok = stn == 0
    print(f"S2 eps={eps:g} prox={prox}: {dt*1e3:8.1f} ms  ({B*K/dt/1e3:.0f} k/s)  failed {int((~ok).sum())} of {ok.size}  "
          f"iters {eng.last_iterations(B)}  |a-a(1e-4)| max {d[ok].max():.2e}  >1e-6: {int((d[ok] > 1e-6).sum())}  >1e-3: {int((d[ok] > 1e-3).sum())}", flush=True)

# use case 10, step 0 (open loop)
o = default_opt(); o["useCaseNum"] = 10
OPT = Settings(o, tree="ABO", N_hor=20); V = SetVehicleParameters("ABO")
for eps, prox in [(1e-4, -1), (0.01, 40), (0.1, 40), (1.0, 40)]:
    BL = Settings_BL(OPT); BL["bl_lp_eps"] = eps; BL["bl_prox_iter"] = prox
    eng = Engine(BL, V, device=0, max_batch=4)
    s_tv = np.asarray(OPT["s_tv"], dtype=np.float64)[:1]; v_tv = np.zeros(1)
    out, _, _, st = eng.ab_step(np.full(1, OPT["s_init"]), np.full(1, OPT["v_init"]), np.full(1, OPT["a_minus1"]), np.zeros(1),
                                s_tv, v_tv, np.zeros(1), want_pred=False)
    print(f"UC10 step0 eps={eps:g} prox={prox}: status {int(st[0])}  a_qp {float(out[OUT['a_qp'], 0]):.9f}  xi_f {float(out[OUT['xi_f'], 0]):.6e}  iters {eng.last_iterations(1)}", flush=True)
