import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from oracle.loader import Oracle
import gpu_qp
OPT, V, _, _ = make_case("ABO", 20)
OPT["Mb"] = np.array([0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 1, 0, 1, 0, 1, 1, 1, 0, 1], dtype=np.int32)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(2, 25, lead["V_TO_2Hz"], seed=9)
OPT["v_init"] = float(sc["v0"][0])
eng = Engine(OPT, V, device=0, max_batch=64)
orc = Oracle(OPT, V)
probs = gpu_qp.fb_problems(orc, OPT, V, sc["s_tv"][:, 0], sc["v_tv"][:, 0], 9)
gpu_qp.run(eng, probs, "FBMB", True)
for p in probs: print(p["qp"])
