"""GPU structured FBMPC kernel against the CPU oracle: golden closed loop (N=20) and S2 scenarios (N=30)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, make_case
from oracle import Oracle
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2

names = ("s", "v", "a", "xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "cost", "DistHor")

def report(tag, tr, st, ref, rst):
    print(tag, "status gpu", int((st != 0).sum()), np.where(st != 0)[0][:10], "oracle", int((rst != 0).sum()), np.where(rst != 0)[0][:10], flush=True)
    ok = (st == 0) & (rst == 0)
    first_bad = np.argmax(~ok) if (~ok).any() else len(ok)
    sl = slice(0, first_bad)
    line = []
    for n in names:
        d = np.abs(tr[sl, OUT[n]] - ref[sl, OUT[n]])
        if n == "cost": d = d / np.maximum(1.0, np.abs(ref[sl, OUT[n]]))
        line.append(f"{n}:{d.max() if d.size else 0:.1e}")
    print("   first", first_bad, "steps:", " ".join(line), flush=True)

for tree in ("ABO", "ORIG"):
    OPT, V, s_tv, v_tv = make_case(tree, 20)
    G = load_golden(f"{tree.lower()}_fbmpc")
    n = int(os.environ.get("NSTEPS", "871"))
    eng = Engine(OPT, V, device=0, max_batch=4)
    B = 2
    stv = np.repeat(s_tv[:n, None], B, 1); vtv = np.repeat(v_tv[:n, None], B, 1)
    t = time.time()
    traj, status = eng.run_fbmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
    eng.synchronize()
    print(tree, "gpu time", time.time() - t, "iters", eng.last_iterations(B) / n)
    tr = traj.cpu().numpy(); st = status.cpu().numpy()
    ref, rst, _ = Oracle(OPT, V).run("fb", n, 0.0, 0.0, 0.0, s_tv[:n].copy(), v_tv[:n].copy())
    report(tree + " golden loop", tr[:, :, 0], st[:, 0], ref, rst)
    print("   vs golden: s %.1e v %.1e Fm(k>=1) %.1e Fb(k>=1) %.1e inst0==inst1 %s" % (
        np.abs(tr[:, OUT["s"], 0] - G["s_opt"][:n]).max(), np.abs(tr[:, OUT["v"], 0] - G["v_opt"][:n]).max(),
        np.abs(tr[1:, OUT["Fm"], 0] - G["Fm_opt"][1:n]).max(), np.abs(tr[1:, OUT["Fb"], 0] - G["Fb_opt"][1:n]).max(),
        np.array_equal(tr[:, :, 0], tr[:, :, 1])), flush=True)

OPT, V, _, _ = make_case("ABO", 30)
lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
B, n = 64, 40
sc = make_s2(B, n, lead["V_TO_2Hz"])
eng = Engine(OPT, V, device=0, max_batch=B)
t = time.time()
traj, status = eng.run_fbmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
eng.synchronize()
print("S2 N=30 gpu time", time.time() - t, "bad", int((status.cpu().numpy() != 0).sum()), "of", B * n)
tr = traj.cpu().numpy(); st = status.cpu().numpy()
orc = Oracle(OPT, V)
for i in range(int(os.environ.get("NORC", "8"))):
    ref, rst, _ = orc.run("fb", n, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
    report(f"S2 inst {i}", tr[:, :, i], st[:, i], ref, rst)
