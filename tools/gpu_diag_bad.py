#!/usr/bin/env python3
"""Diagnose non-zero statuses of the S2 bench workload against the oracle (dev tool)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from oracle import Oracle
N = 30; B = 4096; K = 110
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
st = status.cpu().numpy(); tr = traj.cpu().numpy()
ks, bs = np.nonzero(st)
print("bad:", len(ks), "instances:", sorted(set(bs.tolist()))[:20])
orc = Oracle(OPT, V)
for b in sorted(set(bs.tolist()))[:6]:
    kb = ks[bs == b]
    ref, rst, _ = orc.run("ab", K, 0.0, float(sc["v0"][b]), 0.0, sc["s_tv"][:, b].copy(), sc["v_tv"][:, b].copy())
    print("inst", b, "gpu bad steps", kb.tolist(), "oracle bad steps", np.nonzero(rst)[0].tolist())
    k0 = kb[0]
    for k in (k0 - 1, k0, k0 + 1):
        if 0 <= k < K:
            print("   k", k, "gpu s,v,Fm,xi:", tr[k, [0, 1, 2, 5, 6, 7, 8], b], "\n        orc:", ref[k, [0, 1, 2, 5, 6, 7, 8]])
    print("   gap at k0:", sc["s_tv"][k0, b] - tr[k0, 0, b], "v_tv", sc["v_tv"][k0, b])
