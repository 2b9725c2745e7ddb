import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from oracle import Oracle
N = 30; B = 4096; K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
OPT, V, _, _ = make_case("ABO", N)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(B, K, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
st = status.cpu().numpy(); tr = traj.cpu().numpy()
ks, bs = np.nonzero(st)
print("bad:", [(int(k), int(b), int(st[k, b])) for k, b in zip(ks, bs)][:20])
orc = Oracle(OPT, V)
for b in sorted(set(bs.tolist()))[:4]:
    ref, rst, _ = orc.run("ab", K, 0.0, float(sc["v0"][b]), 0.0, sc["s_tv"][:, b].copy(), sc["v_tv"][:, b].copy())
    k0 = ks[bs == b][0]
    print("inst", b, "oracle bad", np.nonzero(rst)[0].tolist(), "max |dv| before", np.abs(tr[:k0 + 1, 1, b] - ref[:k0 + 1, 1]).max(), "after", np.abs(tr[:, 1, b] - ref[:, 1]).max())
    print("   state at k0:", tr[k0, :9, b], "gap", sc["s_tv"][k0, b] - tr[k0, 0, b], "v_tv", sc["v_tv"][k0, b])
