"""Run-to-run determinism probe of the structured FBMPC kernel (debug aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd._abi import OUT
N, B, n = 30, 4096, 48
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, n, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
runs = []
for r in range(3):
    traj, status = eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"]); eng.synchronize()
    runs.append((traj.cpu().numpy().copy(), status.cpu().numpy().copy()))
t0, s0 = runs[0]
print("nan count", np.isnan(t0).sum(), "status counts", {int(k): int((s0 == k).sum()) for k in np.unique(s0)})
for r in (1, 2):
    t, s = runs[r]
    d = (t != t0) & ~(np.isnan(t) & np.isnan(t0))
    print("run", r, "differing values", d.sum(), "status diffs", (s != s0).sum())
    if d.any():
        idx = np.argwhere(d)
        step, item, inst = idx[0]
        insts = np.unique(idx[:, 2])
        print(" first diff at step", step, "item", item, "inst", inst, t0[step, item, inst], t[step, item, inst], "instances", insts[:20], len(insts))
        for i in insts[:4]:
            fs = np.argwhere(d[:, :, i].any(axis=1))[0, 0]
            print("  inst", i, "first step", fs, "status a/b", s0[max(fs-1,0):fs+2, i], s[max(fs-1,0):fs+2, i], "v0", sc["v0"][i])
            print("   a", t0[fs, :, i]); print("   b", t[fs, :, i])
nn = np.argwhere(np.isnan(t0))
if len(nn):
    insts = np.unique(nn[:, 2])
    print("NaN instances", insts[:30], len(insts))
    for i in insts[:6]:
        fs = nn[nn[:, 2] == i][:, 0].min()
        print(" inst", i, "first NaN step", fs, "status", s0[max(fs - 2, 0):fs + 2, i], "v0", sc["v0"][i], "items", np.unique(nn[(nn[:, 2] == i) & (nn[:, 0] == fs)][:, 1]))
        print("   prev", t0[max(fs - 1, 0), :, i]); print("   at  ", t0[fs, :, i])
