"""Phase timers of the structured FB kernel (library built with EEPACC_EXTRA_FLAGS=-DEEPACC_FBS_TIMING)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine, load_library
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
N, B, K, W = 30, 4096, 200, 200
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, W + K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
lib = load_library()
out = (C.c_ulonglong * 16)()
eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][:W], sc["v_tv"][:W]); eng.synchronize()
lib.eepacc_debug_fbs_prof(out, 1)
eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][W:], sc["v_tv"][W:], resume=True); eng.synchronize()
lib.eepacc_debug_fbs_prof(out, 0)
v = np.array(out[:], dtype=np.float64)
names = {0: "setup: estimators, bounds, scalings, g0, H build", 1: "H inversion", 2: "rows (ba) + state init", 4: "rebuild_and_factor (he_sync, S, P)",
         5: "multipliers + primal + refinement", 6: "find_violation", 7: "step: u = He c, ratio test, events", 8: "final refinement"}
tot = v[:15].sum(); steps = v[15]
print("steps", steps, "ticks per step", tot / steps, "(100 MHz wall clock: %.1f us)" % (tot / steps / 100.0))
for k, nm in names.items():
    print("%-55s %6.1f %%  %8.1f ticks/step" % (nm, 100 * v[k] / tot, v[k] / steps))
print("iters/step", eng.last_iterations(B).mean() / K)
