"""Dev check of the dense QP operator (B3) against the oracle QP on AB and FB dense QPs."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_case, load_golden, golden_step_inputs
from eepacc_mpc_casadi_matlab_amd.engine import Engine
from oracle.loader import Oracle, LoopState
log = open(os.path.join(ROOT, "gpurun_out", "qp.log"), "w")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)

def fb_problems(orc, OPT, V, s_tv, v_tv, n_steps):
    N = OPT["N_hor"]; Ts = OPT["Tvec"][0]
    st = LoopState()
    lm = V["lambda"] * V["m"]
    v0 = OPT.get("v_init", 0.0)
    for k in range(N):
        st.fbA22[k] = 1.0 - 2.0 * OPT["Tvec"][k] * V["zeta_a"] * v0 * v0 / lm
        st.fbD2[k] = OPT["Tvec"][k] / lm * V["zeta_a"] * v0 * v0
    out = []
    s, v, v_prev, a_prev, Fm, Fb, vtvm = 0.0, v0, 5.0, 0.0, 0.0, 0.0, 0.0
    for kk in range(n_steps):
        st.k = kk
        if kk == 0:
            stv, vtv, atv = float(s_tv[0]), 0.0, 0.0
        else:
            s_n, v_n = orc.plant(s, v, Fm, Fb)
            v_prev = v; a_prev = (v_n - v) / Ts; s, v = s_n, v_n
            stv = float(s_tv[kk]); vp = vtvm; vtvm = float(v_tv[kk]); vtv = vtvm; atv = (vtvm - vp) / Ts
        xw = np.array(st.xwarm[:6 * N])
        r = orc.fb_step(st, s, v, v_prev, a_prev, Fm, Fb, kk * Ts, stv, vtv, atv, want_dense=True)
        r["x0"] = xw
        out.append(r)
        Fm, Fb = r["out"][2], r["out"][3]
    return out

def run(eng, probs, tag, use_x0):
    H = np.stack([p["H"] for p in probs]); g = np.stack([p["c"] for p in probs]); A = np.stack([p["G"] for p in probs])
    lb = np.stack([p["lb"] for p in probs]); ub = np.stack([p["ub"] for p in probs])
    x0 = np.stack([p["x0"] for p in probs]) if use_x0 else None
    P(tag, "batch", H.shape, A.shape)
    t0 = time.time()
    x, cost, status = eng.qp_solve_batched(H, g, A, lb, ub, x0=x0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    x = x.cpu().numpy(); status = status.cpu().numpy()
    it = eng.last_iterations(len(probs))
    for i, p in enumerate(probs):
        err = np.abs(x[i] - p["x"]).max()
        P(f"  {tag}[{i}] status {status[i]} (oracle {p['status']}) iters {it[i]} (oracle {p['qp']['iterations']}) max|dx| {err:.3e} |x|max {np.abs(p['x']).max():.3e}")
    P(f"  {tag}: {dt*1e3:.1f} ms for {len(probs)} QPs")

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    OPT, V, s_tv, v_tv = make_case("ABO", 20)
    eng = Engine(OPT, V, device=0, max_batch=64)
    orc = Oracle(OPT, V)
    if which in ("all", "ab"):
        G = load_golden("abo_abmpc")
        probs = []
        for k in (0, 1, 50, 120, 300, 500, 700):
            r = orc.ab_step(**golden_step_inputs(G, s_tv, v_tv, k), want_dense=True)
            probs.append(r)
        run(eng, probs, "AB20", False)
    if which in ("all", "fb"):
        probs = fb_problems(orc, OPT, V, s_tv, v_tv, 10)
        run(eng, probs, "FB20", True)


if __name__ == "__main__":
    main()
