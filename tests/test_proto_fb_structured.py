"""The numpy model of the structured FBMPC solver (tools/proto_fb_structured.py) against the dense oracle: documents
that u = Fm + Fb with eliminated slacks and a pinned friction-brake share w = -Fb solves the reference's (non-convex)
dense QP, on golden states and on a scenario that brakes harder than the motor can regenerate."""
import os
import sys

import numpy as np
import pytest

from conftest import make_case, load_golden, ROOT
from oracle.loader import Oracle, LoopState
from eepacc_mpc_casadi_matlab_amd._abi import OUT
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s2

sys.path.insert(0, os.path.join(ROOT, "tools"))
from proto_fb_structured import FBProblem, StructuredFB  # noqa: E402


def _loop(OPT, V, orc, v0, s_tv, v_tv, n_steps, check):
    """closed loop on the oracle's states; at every step the structured model solves the same QP"""
    N, Ts = OPT["N_hor"], OPT["Tvec"][0]
    lm = V["lambda"] * V["m"]
    st = LoopState(); A22 = np.ones(N); D2 = np.zeros(N)
    for k in range(N):                                   # ABO/RunOpt_FBMPC.m:78-90
        A22[k] = 1 - 2 * OPT["Tvec"][k] * V["zeta_a"] * v0 * v0 / lm
        D2[k] = OPT["Tvec"][k] / lm * V["zeta_a"] * v0 * v0
        st.fbA22[k] = A22[k]; st.fbD2[k] = D2[k]
    prev = None; vtvm = 0.0; t0 = 0.0
    for kk in range(n_steps):
        st.k = kk
        if kk == 0:
            s, v, ap, vprev, Fmp, Fbp = 0.0, v0, 0.0, 5.0, 0.0, 0.0
            stv, vtv, atv = s_tv[0], 0.0, 0.0
        else:
            s, v = orc.plant(*prev)
            vprev, Fmp, Fbp = prev[1], prev[2], prev[3]
            ap = (v - vprev) / Ts; stv = s_tv[kk]; vtvp = vtvm; vtvm = v_tv[kk]; vtv = vtvm; atv = (vtvm - vtvp) / Ts
        r = orc.fb_step(st, s, v, vprev, ap, Fmp, Fbp, t0, stv, vtv, atv, want_dense=True)
        prob = FBProblem(OPT, V, s, v, ap, t0, stv, vtv, atv, A22, D2, kk)
        qp = StructuredFB(prob)
        status = qp.solve()
        if prob.infeasible_const:
            status = 1
        check(kk, r, prob, qp, status)
        o = r["out"]; prev = (s, v, o[OUT["Fm"]], o[OUT["Fb"]]); t0 += Ts


def _dense_x(prob, qp):
    u, w, xi = qp.solution()
    N = prob.N
    return prob.dense_x(u, w, {(kind, k): xi.get((kind, k), 0.0) for kind in "vhsf" for k in range(N)}), u, w


@pytest.mark.parametrize("tree", ["ABO", "ORIG"])
def test_structured_fb_equals_dense_qp_on_golden_states(tree):
    OPT, V, s_tv, v_tv = make_case(tree, 20)
    orc = Oracle(OPT, V)
    seen = []

    def check(kk, r, prob, qp, status):
        assert status == 0 and r["status"] == 0
        x, u, w = _dense_x(prob, qp)
        xr = r["x"]
        if kk > 0:          # k = 0: standstill, the force split is a degenerate face
            assert np.abs(x[0::6] + x[1::6] - xr[0::6] - xr[1::6]).max() < 1e-6
            assert np.abs(x[1::6] - xr[1::6]).max() < 1e-6
        for i in range(2, 6):
            assert np.abs(x[i::6] - xr[i::6]).max() < 1e-8
        H, c = r["H"], r["c"]
        assert abs((0.5 * x @ H @ x + c @ x) - (0.5 * xr @ H @ xr + c @ xr)) < 1e-9 * abs(0.5 * xr @ H @ xr + c @ xr)
        seen.append(kk)

    _loop(OPT, V, orc, 0.0, s_tv, v_tv, 40, check)
    assert len(seen) == 40


def test_structured_fb_hard_braking():
    """S2 instance 16: the first steps brake below the regeneration limit (torque / rear-axle rows become the pivot
    of w, rank-2 updates of the Hessian); dense feasibility and objective of the structured solution are checked
    against the oracle's dense QP."""
    OPT, V, _, _ = make_case("ABO", 20)
    lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
    sc = make_s2(17, 8, lead["V_TO_2Hz"])
    orc = Oracle(OPT, V)
    used_w = []

    def check(kk, r, prob, qp, status):
        assert status == 0 and r["status"] == 0, (kk, status, r["status"])
        x, u, w = _dense_x(prob, qp)
        xr = r["x"]
        Gx = r["G"] @ x
        assert max(np.max(Gx - r["ub"]), np.max(r["lb"] - Gx)) < 1e-7            # feasible in the dense QP
        H, c = r["H"], r["c"]
        cp, co = 0.5 * x @ H @ x + c @ x, 0.5 * xr @ H @ xr + c @ xr
        assert cp <= co + 1e-9 * abs(co)                                         # at least as good as the oracle's point
        assert abs(cp - co) < 1e-9 * abs(co)
        assert np.abs(x[0::6] + x[1::6] - xr[0::6] - xr[1::6]).max() < 1e-4
        used_w.append(float(w.max()))

    _loop(OPT, V, orc, float(sc["v0"][16]), sc["s_tv"][:, 16], sc["v_tv"][:, 16], 6, check)
    assert max(used_w) > 100.0           # the friction brake really was in use


def test_structured_fb_emergency_first_step():
    """S2 instance 33 (N = 30), step 0: the whole friction brake is needed (Fb = -1e4 N) together with a large xi_f.
    w sits on its upper bound: the bound row becomes w's pivot and rows that contain w may then define xi_f.  Dense
    feasibility and objective against the oracle's dense QP."""
    OPT, V, _, _ = make_case("ABO", 30)
    lead = np.load(os.path.join(ROOT, "tests", "golden", "lead_TO01_EAD.npz"))
    sc = make_s2(34, 2, lead["V_TO_2Hz"])
    orc = Oracle(OPT, V)
    seen = []

    def check(kk, r, prob, qp, status):
        assert status == 0 and r["status"] == 0, (kk, status, r["status"])
        x, u, w = _dense_x(prob, qp)
        xr = r["x"]
        Gx = r["G"] @ x
        assert max(np.max(Gx - r["ub"]), np.max(r["lb"] - Gx)) < 1e-6
        H, c = r["H"], r["c"]
        cp, co = 0.5 * x @ H @ x + c @ x, 0.5 * xr @ H @ xr + c @ xr
        assert abs(cp - co) < 1e-9 * abs(co)
        assert abs(x[0] + x[1] - xr[0] - xr[1]) < 1e-4 and abs(x[1] + 1e4) < 1e-6 and abs(xr[1] + 1e4) < 1e-6
        assert qp.wst[0]["P"] and prob.rows[qp.wst[0]["pivot"]]["name"] == "fb_lo"
        seen.append(kk)

    _loop(OPT, V, orc, float(sc["v0"][33]), sc["s_tv"][:, 33], sc["v_tv"][:, 33], 1, check)
    assert seen == [0]
