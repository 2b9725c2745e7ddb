"""Per-step comparison of selected S2 instances (structured FB kernel vs oracle)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_case
from oracle import Oracle
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2
N = int(sys.argv[1]); inst = [int(x) for x in sys.argv[2].split(",")]; n = int(sys.argv[3])
import json
over = json.loads(os.environ.get("OPT_OVERRIDE", "{}"))
OPT, V, _, _ = make_case("ABO", N, **over)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
sc = make_s2(max(inst) + 1, n, lead["V_TO_2Hz"], seed=int(os.environ.get("SEED", "1234")))
if os.environ.get("GAPS"):
    sc["s_tv"] = sc["s_tv"] + np.array([float(x) for x in os.environ["GAPS"].split(",")])[None, :]
eng = Engine(OPT, V, device=0, max_batch=len(inst))
idx = np.array(inst)
traj, status = eng.run_fbmpc(sc["s0"][idx], sc["v0"][idx], sc["a_minus1"][idx], sc["s_tv"][:, idx].copy(), sc["v_tv"][:, idx].copy())
eng.synchronize()
tr = traj.cpu().numpy(); st = status.cpu().numpy()
orc = Oracle(OPT, V)
for j, i in enumerate(inst):
    ref, rst, it = orc.run("fb", n, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
    for k in range(n):
        d = {nm: abs(tr[k, OUT[nm], j] - ref[k, OUT[nm]]) for nm in ("s", "v", "Fm", "Fb", "xi_v", "xi_h", "xi_s", "xi_f")}
        d["cost"] = abs(tr[k, OUT["cost"], j] - ref[k, OUT["cost"]]) / max(1.0, abs(ref[k, OUT["cost"]]))
        flag = "  <<<" if (max(d["s"], d["v"]) > 1e-8 or d["Fm"] > 1e-5 or st[k, j] != rst[k]) else ""
        print(i, k, "st", st[k, j], rst[k], " ".join(f"{a}:{b:.1e}" for a, b in d.items()),
              "Fm %.3f Fb %.3f | orc %.3f %.3f v %.4f" % (tr[k, OUT["Fm"], j], tr[k, OUT["Fb"], j], ref[k, OUT["Fm"]], ref[k, OUT["Fb"]], ref[k, OUT["v"]]), flag, flush=True)
