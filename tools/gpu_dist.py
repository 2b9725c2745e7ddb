import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
N = 30; B = 4096; W = 20; K = 200
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, W + K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][:W], sc["v_tv"][:W]); torch.cuda.synchronize()
t = time.perf_counter()
eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][W:], sc["v_tv"][W:], resume=True); torch.cuda.synchronize()
dt = time.perf_counter() - t
x = eng.last_iterations(B).astype(np.float64)
print("kernel wall %.1f ms" % (dt * 1e3), "per-instance value: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f sum/2048 %.1f" % (x.mean(), np.percentile(x, 50), np.percentile(x, 90), np.percentile(x, 99), x.max(), x.sum() / 2048))
