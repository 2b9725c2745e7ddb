"""Baseline controller on the bench's S2 scenarios: finds the failed steps of a closed loop and replays some of them as
single open-loop steps (with a library built with -DEEPACC_BL_TRACE the replay prints its event trace)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.settings import Settings_BL
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
n_replay = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, B, K = 30, 4096, 25
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, K, lead["V_TO_2Hz"], first_instance=0)
OPT, V, _, _ = make_case("ABO", N)
BL = Settings_BL(OPT); BL["bl_lp_eps"] = eps
Ts = float(BL["Tvec"][0])
eng = Engine(BL, V, device=0, max_batch=B)
traj, st = eng.run_abmpc(*[torch.as_tensor(sc[k], device="cuda:0") for k in ("s0", "v0", "a_minus1", "s_tv", "v_tv")])
tr = traj.cpu().numpy(); stn = st.cpu().numpy()
ks, bs = np.nonzero(stn)
print("failed", len(ks), "steps; by step:", np.bincount(ks, minlength=K), flush=True)
print("speeds at the failed steps: min %.3g median %.3g max %.3g" % tuple(np.percentile(tr[ks, OUT["v"], bs], [0, 50, 100])))
eng1 = Engine(BL, V, device=0, max_batch=4)
for k, b in list(zip(ks, bs))[:n_replay]:
    s, v = tr[k, OUT["s"], b], tr[k, OUT["v"], b]
    a_prev = sc["a_minus1"][b] if k == 0 else (tr[k, OUT["v"], b] - tr[k - 1, OUT["v"], b]) / Ts
    vm = sc["v_tv"][k, b] if k > 0 else 0.0
    vm1 = (sc["v_tv"][k - 1, b] if k > 1 else 0.0)
    atv = (vm - vm1) / Ts if k > 0 else 0.0
    print(f"--- replay step {k} instance {b}: s {s:.6f} v {v:.9f} a_prev {a_prev:.6f} s_tv {sc['s_tv'][k, b]:.6f} v_tv {vm:.6f}", flush=True)
    out, sp, vp, s1 = eng1.ab_step(np.full(1, s), np.full(1, v), np.full(1, a_prev), np.full(1, k * Ts), np.full(1, sc["s_tv"][k, b]),
                                   np.full(1, vm), np.full(1, atv), want_pred=True)
    torch.cuda.synchronize()
    print("replay status", int(s1[0]), "a_qp", float(out[OUT["a_qp"], 0]), "closed-loop a_qp", tr[k, OUT["a_qp"], b], "iters", eng1.last_iterations(1), flush=True)
    print("v_pred", np.array2string(vp[:, 0].cpu().numpy(), precision=4), flush=True)
