"""Phase timing of the ABMPC kernel (needs a build with -DEEPACC_AB_TIMING)."""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine, load_library
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
log = open(os.path.join(ROOT, "gpurun_out", "ab_prof.log"), "w")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
N, B, n = 30, 4096, 220
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, n, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
lib = load_library()
names = ["rebuild+factor", "multipliers", "refine", "warm repair", "find violation", "step/apply", "setup", "outputs", "solve total", " he_sync", " list", " S build", " inversion", "passes"]
why_names = ["first pass", "warm repair", "EV_CAP", "COMPL/CAPIN/DROPH", "bound add", "duplicate", "cap reset", "cold", "add with m=0"]
prof = (C.c_ulonglong * 24)()
mvals = []
stv = torch.as_tensor(sc["s_tv"], device="cuda"); vtv = torch.as_tensor(sc["v_tv"], device="cuda")
eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[:20], vtv[:20]); torch.cuda.synchronize()
lib.eepacc_debug_ab_prof(prof, 1)
t0 = time.time()
eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], stv[20:], vtv[20:], resume=True); torch.cuda.synchronize()
dt = time.time() - t0
lib.eepacc_debug_ab_prof(prof, 1)
tot = sum(prof[i] for i in (6, 7, 8))
P(f"200 steps x {B}: {dt*1e3:.1f} ms (instrumented)")
for i, nm in enumerate(names):
    P(f"  {nm:16s} {prof[i] / 100.0 / (B * 200):8.3f} us/step  {100.0 * prof[i] / tot:5.1f} %")
P("passes/step", prof[13] / (B * 200.0))
for i, nm in enumerate(why_names):
    P(f"  full rebuilds/step because {nm:20s} {prof[14 + i] / (B * 200.0):.3f}")
P("carried passes/step", (prof[23] % 1000) / (B * 200.0), "(low digits only)  raw", prof[23] / (B * 200.0))
P("iterations/step", eng.last_iterations(B).mean() / 200)
