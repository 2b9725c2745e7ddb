"""Phase timing (library built with -DEEPACC_AB_TIMING) of ONE S2 instance's closed loop: where a hard instance's passes go."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine, load_library
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
inst = int(sys.argv[1]); W = 5; K = 20
OPT, V, _, _ = make_case("ABO", 30)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(1, W + K, lead["V_TO_2Hz"], first_instance=inst)
eng = Engine(OPT, V, device=0, max_batch=1)
lib = load_library()
names = ["rebuild+factor", "multipliers", "refine", "warm repair", "find violation", "step/apply", "setup", "outputs", "solve total", " he_sync", " list", " S build", " inversion", "passes"]
why_names = ["first pass", "warm repair", "EV_CAP", "COMPL/CAPIN/DROPH", "bound add", "duplicate", "cap reset", "cold", "add with m=0"]
prof = (C.c_ulonglong * 24)()
stv = torch.as_tensor(sc["s_tv"], device="cuda"); vtv = torch.as_tensor(sc["v_tv"], device="cuda")
eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[:W], vtv[:W]); torch.cuda.synchronize()
lib.eepacc_debug_ab_prof(prof, 1)
eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[W:], vtv[W:], resume=True); torch.cuda.synchronize()
lib.eepacc_debug_ab_prof(prof, 1)
tot = sum(prof[i] for i in (6, 7, 8))
print(f"instance {inst}: {K} steps, {tot / 100.0 / K:.1f} us/step (one wave alone on the GPU)")
for i, nm in enumerate(names):
    print(f"  {nm:16s} {prof[i] / 100.0 / K:8.2f} us/step  {100.0 * prof[i] / tot:5.1f} %")
print("passes/step", prof[13] / K)
for i, nm in enumerate(why_names):
    print(f"  full rebuilds/step because {nm:20s} {prof[14 + i] / K:.2f}")
print("iterations/step", float(np.mean(eng.last_iterations(1))) / K)
