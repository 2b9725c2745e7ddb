"""CPU oracle of the full-route nonlinear problem of RunOpt_NLP (TEST INFRASTRUCTURE, numpy).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Two parts:

1. ``NlpProblem`` -- a restatement of the NLP that ABO/RunOpt_NLP.m builds (ORIG/RunOpt_NLP.m is the
   same file): lookup tables (:63-184), model and running cost (:204-245), RK4 x 4 integrator
   (:262-278), the multiple-shooting variable layout (:296-365), every constraint row (:367-501) with
   its bounds, and the post-processing (:545-605).  ``eval_reference_form(z)`` returns J, g, lbg, ubg,
   lbz, ubz in the reference's own ordering, so it can be checked against the saved IPOPT solution
   (tests/golden/{abo,orig}_nlp.npz): continuity defects, theta / jerk equalities, all inequality rows
   and the slack complementarity pin the integrator, the jerk definition and every lookup table.
   CasADi's ``interpolant('LUT','linear')`` is a dependency that is not in the tree as source
   (CasADi 3.6.3 binaries only); it is restated as linear interpolation with linear extrapolation from
   the end segments (CasADi's documented behaviour for the 'linear' plugin).

2. ``solve`` -- a structured primal-dual interior-point method (stage-wise Riccati recursion, exact
   Lagrangian Hessian with inertia regularisation, l1 merit line search).  IPOPT's source is not in
   the reference tree either; parity with the reference is therefore *objective level* (SURVEY.md
   section 8f rank 2): J of the saved solution, evaluated by part 1, against J of this solver and of
   the HIP kernel (which implements the same method stage by stage, so the two are also compared
   iterate for iterate on small problems).

Stage form used by the solver (and by csrc/eepacc_nlp.hip): state chi = (s, v, p, j), control
u = (Fm, Fb, xi_v, xi_h, xi_s, xi_f); p_k is the acceleration at node k under the *previous* force
(the bracket of the jerk equality, RunOpt_NLP.m:371-375), so that j_{k+1} = (p_{k+1} - p_k)/Ts is a
state recursion and the problem is Markov; theta_k = slopeLookup(s_k) is substituted (:362-369).
"""
from __future__ import annotations

import math
from typing import Any, Dict

import numpy as np

NX, NU, NY = 4, 6, 10           # chi, u, y = (chi', u)
IS, IV, IP, IJ, IFM, IFB, IXV, IXH, IXS, IXF = range(10)
A_HWP, T_HWP = 2.0, 2.0
G_HWP = -0.0246 * T_HWP + 0.010819                                   # RunOpt_NLP.m:494-496
ISO_V = np.array([0.0, 5.0, 20.0, 25.0])                             # :79-84
ISO_AMIN = np.array([-4.0, -4.0, -2.0, -2.0])
ISO_AMAX = np.array([5.0, 5.0, 3.5, 3.5])
ISO_JMAG = np.array([5.0, 5.0, 2.5, 2.5])


from .nlp_tables import lookup as pwa, build_tables   # the checker's own restatement of RunOpt_NLP.m:63-184 (no product code)


def pwa_smooth(x, xs, ys, eps):
    """The lookup convolved with a box of half-width eps (eps = 0: the lookup itself): value, slope, curvature.
    With F the antiderivative of the piecewise-linear f:  f_eps = (F(x+eps) - F(x-eps)) / (2 eps), C^1 with bounded
    curvature.  Used by the solver's graduated smoothing only; the problem that is pinned and reported is eps = 0."""
    if eps <= 0.0:
        v, sl = pwa(x, xs, ys)
        return v, sl, np.zeros_like(v)
    xs = np.asarray(xs, dtype=np.float64)
    ys = np.asarray(ys, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    dxs = np.diff(xs)
    seg_sl = np.where(dxs > 0, np.diff(ys) / np.where(dxs > 0, dxs, 1.0), 0.0)
    C = np.concatenate([[0.0], np.cumsum(0.5 * (ys[1:] + ys[:-1]) * dxs)])

    def F_f_df(z):
        i = np.clip(np.searchsorted(xs, z, side="right") - 1, 0, len(xs) - 2)
        d = z - xs[i]
        return C[i] + ys[i] * d + 0.5 * seg_sl[i] * d * d, ys[i] + seg_sl[i] * d, seg_sl[i]
    Fp, fp, dp = F_f_df(x + eps)
    Fm, fm, dm = F_f_df(x - eps)
    return (Fp - Fm) / (2 * eps), (fp - fm) / (2 * eps), (dp - dm) / (2 * eps)


# ----------------------------------------------------------------------------------------------
# second-order jets over (s_k, v_k, Fm_k, F_k)
# ----------------------------------------------------------------------------------------------
class Jet:
    __slots__ = ("v", "g", "h")

    def __init__(self, v, g, h):
        self.v, self.g, self.h = v, g, h

    @staticmethod
    def const(c, n):
        return Jet(np.full(n, float(c)), np.zeros((4, n)), np.zeros((4, 4, n)))

    @staticmethod
    def var(x, i):
        n = len(x)
        g = np.zeros((4, n))
        g[i] = 1.0
        return Jet(np.array(x, dtype=np.float64), g, np.zeros((4, 4, n)))

    def __add__(self, o):
        if isinstance(o, Jet):
            return Jet(self.v + o.v, self.g + o.g, self.h + o.h)
        return Jet(self.v + o, self.g, self.h)
    __radd__ = __add__

    def __neg__(self):
        return Jet(-self.v, -self.g, -self.h)

    def __sub__(self, o):
        return self + (-o)

    def __rsub__(self, o):
        return (-self) + o

    def __mul__(self, o):
        if isinstance(o, Jet):
            gg = self.g[:, None, :] * o.g[None, :, :]
            return Jet(self.v * o.v, self.v * o.g + o.v * self.g,
                       self.v * o.h + o.v * self.h + gg + gg.transpose(1, 0, 2))
        return Jet(self.v * o, self.g * o, self.h * o)
    __rmul__ = __mul__

    def fn(self, f, df, d2f):
        """Scalar function of a jet (chain rule)."""
        return Jet(f, df * self.g, df * self.h + d2f * self.g[:, None, :] * self.g[None, :, :])


class NlpProblem:
    """Data + model of one route.  ``s_tv`` is the lead trace of Main.m:88 (length >= N)."""

    def __init__(self, OPT: Dict[str, Any], V: Dict[str, float], s_tv: np.ndarray):
        self.OPT, self.V = OPT, V
        self.T = build_tables(OPT)
        self.N = self.T["N"]
        self.Ts = float(OPT["Ts"])
        self.W = np.asarray(OPT["W_NLP"], float)
        if OPT.get("useFifthOrderFit_NLP", True):
            self.b = np.asarray(OPT["b_fifthOrder"], float)
        else:
            self.b = np.concatenate([np.asarray(OPT["b_quadr"], float), np.zeros(15)])
        self.s_tv = np.asarray(s_tv, float)[: self.N].copy()
        self.s0, self.v0 = float(OPT["s_init"]), float(OPT["v_init"])
        self.s_goal = float(OPT["s_goal"])
        self.h_min, self.tau_min = float(OPT["h_min"]), float(OPT["tau_min"])
        self.alpha = float(OPT["alpha_TTL"])
        self.Fm_min = -V["phi"] * V["T_m_max"] / V["eta_TF"]                               # :194-195
        self.Fm_max = V["phi"] * V["T_m_max"] * V["eta_TF"]
        self.n_tl = self.T["tl_s"].shape[0]
        self.n_rows = 17 + 2 * self.n_tl + 10 + (1 if math.isfinite(self.s_goal) else 0)
        self.eps_s = self.eps_v = 0.0        # graduated smoothing of the lookups in position / speed (solver only)

    # ---- model pieces --------------------------------------------------------------------------
    def theta(self, s):
        if self.T["flat"]:
            return np.zeros_like(np.asarray(s, float)), np.zeros_like(np.asarray(s, float))
        return pwa(s, *self.T["slope"])

    def drag(self, v, th):
        V = self.V
        return V["zeta_a"] * v * v + V["c_r"] * V["m"] * V["g"] * np.cos(th) + V["m"] * V["g"] * np.sin(th)

    def p_bat(self, Fm, rpm):
        """RunOpt_NLP.m:226-231 (plain floats or jets)."""
        b = self.b
        F2, r2 = Fm * Fm, rpm * rpm
        F3, r3 = F2 * Fm, r2 * rpm
        F4, r4 = F2 * F2, r2 * r2
        return (b[0] + b[1] * Fm + b[2] * rpm + b[3] * F2 + b[4] * (Fm * rpm) + b[5] * r2
                + b[6] * F3 + b[7] * (F2 * rpm) + b[8] * (Fm * r2) + b[9] * r3 + b[10] * F4
                + b[11] * (F3 * rpm) + b[12] * (F2 * r2) + b[13] * (Fm * r3) + b[14] * r4
                + b[15] * (F4 * Fm) + b[16] * (F4 * rpm) + b[17] * (F3 * r2) + b[18] * (F2 * r3)
                + b[19] * (Fm * r4) + b[20] * (r4 * rpm))

    def rk4(self, s, v, Fm, F, cth, sth):
        """F = RK4 x 4 of (xdot, L) over one interval (:262-278), theta and j frozen (xdot(3:4)=0).
        Returns (s', v', integral of w_P*P_bat + w_a*a^2).  Works on floats and on jets."""
        V = self.V
        lm = V["lambda"] * V["m"]
        grav = V["c_r"] * V["m"] * V["g"] * cth + V["m"] * V["g"] * sth
        kr = (30.0 / math.pi) * V["phi"]

        def f(vv):
            a = (F - V["zeta_a"] * (vv * vv) - grav) * (1.0 / lm)
            L = self.W[0] * self.p_bat(Fm, vv * kr) + self.W[1] * (a * a)
            return a, L
        DT = self.Ts / 4
        q = 0.0
        for _ in range(4):
            a1, l1 = f(v)
            v2 = v + (DT / 2) * a1
            a2, l2 = f(v2)
            v3 = v + (DT / 2) * a2
            a3, l3 = f(v3)
            v4 = v + DT * a3
            a4, l4 = f(v4)
            s = s + (DT / 6) * (v + 2 * v2 + 2 * v3 + v4)
            q = q + (DT / 6) * (l1 + 2 * l2 + 2 * l3 + l4)
            v = v + (DT / 6) * (a1 + 2 * a2 + 2 * a3 + a4)
        return s, v, q

    # ---- the reference's own form (pin) ---------------------------------------------------------
    def eval_reference_form(self, s, v, th, j, U):
        """J and the constraint rows in the order of RunOpt_NLP.m:336-501 (multiple shooting).
        s, v, th, j: [N+1]; U: [N][6].  Returns dict(J, eq=[N][4] (continuity s, v, theta, jerk),
        ineq=[N][n_ineq] (each row as  value - bound  in '<= 0' orientation))."""
        V, N, Ts = self.V, self.N, self.Ts
        lm = V["lambda"] * V["m"]
        Fm, Fb = U[:, 0], U[:, 1]
        xv, xh, xs, xf = U[:, 2], U[:, 3], U[:, 4], U[:, 5]
        F = Fm + Fb
        s1, v1, q = self.rk4(s[:-1], v[:-1], Fm, F, np.cos(th[:-1]), np.sin(th[:-1]))
        W = self.W
        J = np.sum(q + Ts * (W[2] * j[:-1] ** 2 + W[3] * xv + W[4] * (xh ** 2 + 1e2 * xh) + W[5] * xs + W[6] * xf))
        sk, vk, tk, jk = s[1:], v[1:], th[1:], j[1:]
        eq = np.zeros((N, 4))
        eq[:, 0] = s1 - sk
        eq[:, 1] = v1 - vk
        eq[:, 2] = tk - self.theta(sk)[0]
        Fp = np.concatenate([[0.0], F[:-1]])
        eq[:, 3] = jk - (F - self.drag(vk, tk) - Fp + self.drag(v[:-1], th[:-1])) / (lm * Ts)
        a = (F - self.drag(vk, tk)) / lm
        mg = V["m"] * V["g"]
        rows = []
        rows.append(-(Fm * vk + V["P_m_max"] / V["eta_TF"] + xf))
        rows.append(Fm * vk - V["P_m_max"] * V["eta_TF"] - xf)
        rows.append(-(F + V["mu"] * mg * np.cos(tk) + xf))
        rows.append(F - V["mu"] * mg * np.cos(tk) - xf)
        rear = V["h_g"] * V["lambda"] * a + V["h_g"] * V["zeta_a"] / V["m"] * vk ** 2 \
            + V["g"] * (V["L_f"] * np.cos(tk) + V["h_g"] * np.sin(tk))
        rows.append(-(V["L"] / (V["mu"] * V["m"]) * Fm + rear + xf))
        rows.append(V["L"] / (V["mu"] * V["m"]) * Fm - rear - xf)
        rows.append(-(a - pwa(vk, ISO_V, ISO_AMIN)[0] + xf))
        rows.append(a - pwa(vk, ISO_V, ISO_AMAX)[0] - xf)
        jm = pwa(vk, ISO_V, ISO_JMAG)[0]
        rows.append(-(jk + jm + xf))
        rows.append(jk - jm - xf)
        rows.append(vk - pwa(sk, *self.T["vlim"])[0] - xf)
        with np.errstate(divide="ignore"):
            rows.append(vk - self.alpha * np.abs(pwa(sk, *self.T["curv"])[0]) ** (-1.0 / 3.0) - xf)
        rows.append(vk - pwa(sk, *self.T["stop"])[0] - xs)
        for i in range(self.n_tl):
            tv = pwa(sk, self.T["tl_s"][i], self.T["tl_v"])[0]
            tt = self.T["tl_state"][i]
            rows.append(vk - tv - tt - xs)
            rows.append(-(vk + tv + 1e3 - 10 - tt + xs))
        rows.append(-(vk - pwa(sk, *self.T["vinc"])[0] + xv))
        rows.append(sk - (self.s_tv - self.h_min))
        rows.append(sk + self.tau_min * vk - xs - self.s_tv)
        rows.append(sk + vk * T_HWP + vk ** 2 * G_HWP - xh - (self.s_tv - A_HWP))
        rows += [self.Fm_min - Fm, Fm - self.Fm_max, Fb, -xv, -xh, -xs, -xf, -sk, -vk, vk - V["v_max"]]
        if math.isfinite(self.s_goal):
            rows.append(sk - self.s_goal)
        return dict(J=float(J), eq=eq, ineq=np.stack(rows, axis=1), s_end=s1, v_end=v1)

    def postprocess(self, v_opt, Fm_opt):
        """RunOpt_NLP.m:545-553: rpm, P (always the fifth-order surface), E, a, Tm."""
        V = self.V
        rpm = (30 / math.pi) * v_opt[:-1] * V["phi"]
        b5 = np.asarray(self.OPT["b_fifthOrder"], float)
        keep = self.b
        self.b = b5
        P = self.p_bat(Fm_opt, rpm)
        self.b = keep
        E = self.Ts * np.cumsum(P)
        a = np.diff(v_opt) / self.Ts
        Tm = Fm_opt / V["phi"] / (V["eta_TF"] ** np.sign(Fm_opt))
        return dict(rpm_opt=rpm, P_opt=P, E_opt=E, a_opt=a, Tm_opt=Tm)


# ----------------------------------------------------------------------------------------------
# structured interior-point solver (the method csrc/eepacc_nlp.hip implements)
# ----------------------------------------------------------------------------------------------
_JMAP = ((0, IS), (1, IV), (2, IFM), (3, IFM), (3, IFB))      # jet variable -> (chi, u) coordinate(s)


def _jet_grad10(J, n):
    g = np.zeros((n, NY))
    for a, i in _JMAP:
        g[:, i] += J.g[a]
    return g


def _jet_hess10(J, n):
    H = np.zeros((n, NY, NY))
    for a, i in _JMAP:
        for b, j in _JMAP:
            H[:, i, j] += J.h[a, b]
    return H


class NlpOptions:
    def __init__(self, **kw):
        self.obj_scale = 1e-5        # sigma: scaled objective = sigma * J
        self.mu_init = 1.0
        self.mu_min = 1e-9
        self.kappa_eps = 10.0
        self.kappa_mu = 0.2
        self.kappa_sigma = 1e10      # multipliers are kept within this factor of mu / t: the slacks follow their rows (primal),
                                     # so the barrier curvature the merit sees is mu / t^2
        self.theta_mu = 1.5
        self.tol = 1e-7
        self.tau_min = 0.99
        self.max_iter = 600
        self.max_ls = 4
        self.reg_first = 1e-4
        self.reg_max = 1e8
        self.verbose = False
        self.primal = False          # True: rows that hold use lam = mu / t (Newton on the primal barrier function)
        self.__dict__.update(kw)


def _rows(P: NlpProblem, chi1, u, k_idx):
    """All inequality rows of the stages (orientation r <= 0): values [n][R], gradients [n][R][10] in
    y = (chi_{k+1}, u_k), and curvature entries as a list of (row, i, j, values[n])."""
    V = P.V
    n = chi1.shape[0]
    s, v, p, j = chi1[:, 0], chi1[:, 1], chi1[:, 2], chi1[:, 3]
    Fm, Fb, xv, xh, xs, xf = (u[:, i] for i in range(6))
    F = Fm + Fb
    th, dth = P.theta(s)
    cth, sth = np.cos(th), np.sin(th)
    mg = V["m"] * V["g"]
    R = P.n_rows
    r = np.zeros((n, R))
    G = np.zeros((n, R, NY))
    curv = []
    i = 0
    r[:, i] = -(Fm * v + V["P_m_max"] / V["eta_TF"] + xf); G[:, i, IV] = -Fm; G[:, i, IFM] = -v; G[:, i, IXF] = -1
    curv.append((i, IV, IFM, -np.ones(n))); i += 1
    r[:, i] = Fm * v - V["P_m_max"] * V["eta_TF"] - xf; G[:, i, IV] = Fm; G[:, i, IFM] = v; G[:, i, IXF] = -1
    curv.append((i, IV, IFM, np.ones(n))); i += 1
    r[:, i] = -(F + V["mu"] * mg * cth + xf); G[:, i, IFM] = -1; G[:, i, IFB] = -1
    G[:, i, IS] = V["mu"] * mg * sth * dth; G[:, i, IXF] = -1; i += 1
    r[:, i] = F - V["mu"] * mg * cth - xf; G[:, i, IFM] = 1; G[:, i, IFB] = 1
    G[:, i, IS] = V["mu"] * mg * sth * dth; G[:, i, IXF] = -1; i += 1
    kF = V["L"] / (V["mu"] * V["m"])
    kz = V["h_g"] * V["zeta_a"] / V["m"]
    rear = V["h_g"] * V["lambda"] * p + kz * v * v + V["g"] * (V["L_f"] * cth + V["h_g"] * sth)
    drear_s = V["g"] * (-V["L_f"] * sth + V["h_g"] * cth) * dth
    r[:, i] = -(kF * Fm + rear + xf); G[:, i, IFM] = -kF; G[:, i, IP] = -V["h_g"] * V["lambda"]
    G[:, i, IV] = -2 * kz * v; G[:, i, IS] = -drear_s; G[:, i, IXF] = -1
    curv.append((i, IV, IV, np.full(n, -2 * kz))); i += 1
    r[:, i] = kF * Fm - rear - xf; G[:, i, IFM] = kF; G[:, i, IP] = -V["h_g"] * V["lambda"]
    G[:, i, IV] = -2 * kz * v; G[:, i, IS] = -drear_s; G[:, i, IXF] = -1
    curv.append((i, IV, IV, np.full(n, -2 * kz))); i += 1
    es, evv = P.eps_s, P.eps_v
    amin, damin, camin = pwa_smooth(v, ISO_V, ISO_AMIN, evv)
    amax, damax, camax = pwa_smooth(v, ISO_V, ISO_AMAX, evv)
    jm, djm, cjm = pwa_smooth(v, ISO_V, ISO_JMAG, evv)
    r[:, i] = -(p - amin + xf); G[:, i, IP] = -1; G[:, i, IV] = damin; G[:, i, IXF] = -1
    curv.append((i, IV, IV, camin)); i += 1
    r[:, i] = p - amax - xf; G[:, i, IP] = 1; G[:, i, IV] = -damax; G[:, i, IXF] = -1
    curv.append((i, IV, IV, -camax)); i += 1
    r[:, i] = -(j + jm + xf); G[:, i, IJ] = -1; G[:, i, IV] = -djm; G[:, i, IXF] = -1
    curv.append((i, IV, IV, -cjm)); i += 1
    r[:, i] = j - jm - xf; G[:, i, IJ] = 1; G[:, i, IV] = -djm; G[:, i, IXF] = -1
    curv.append((i, IV, IV, -cjm)); i += 1
    vl, dvl, cvl = pwa_smooth(s, *P.T["vlim"], es)
    r[:, i] = v - vl - xf; G[:, i, IV] = 1; G[:, i, IS] = -dvl; G[:, i, IXF] = -1
    curv.append((i, IS, IS, -cvl)); i += 1
    c, dc, _ = pwa_smooth(s, *P.T["curv"], es)
    ac = np.maximum(np.abs(c), 1e-300)
    r[:, i] = v - P.alpha * ac ** (-1.0 / 3.0) - xf; G[:, i, IV] = 1
    G[:, i, IS] = P.alpha / 3.0 * ac ** (-4.0 / 3.0) * np.sign(c) * dc; G[:, i, IXF] = -1; i += 1
    sv, dsv, csv_ = pwa_smooth(s, *P.T["stop"], es)
    r[:, i] = v - sv - xs; G[:, i, IV] = 1; G[:, i, IS] = -dsv; G[:, i, IXS] = -1
    curv.append((i, IS, IS, -csv_)); i += 1
    for t in range(P.n_tl):
        tv, dtv, ctv = pwa_smooth(s, P.T["tl_s"][t], P.T["tl_v"], es)
        tt = P.T["tl_state"][t][k_idx]
        r[:, i] = v - tv - tt - xs; G[:, i, IV] = 1; G[:, i, IS] = -dtv; G[:, i, IXS] = -1
        curv.append((i, IS, IS, -ctv)); i += 1
        r[:, i] = -(v + tv + 1e3 - 10 - tt + xs); G[:, i, IV] = -1; G[:, i, IS] = -dtv; G[:, i, IXS] = -1
        curv.append((i, IS, IS, -ctv)); i += 1
    vi, dvi, cvi = pwa_smooth(s, *P.T["vinc"], es)
    r[:, i] = -(v - vi + xv); G[:, i, IV] = -1; G[:, i, IS] = dvi; G[:, i, IXV] = -1
    curv.append((i, IS, IS, cvi)); i += 1
    stv = P.s_tv[k_idx]
    r[:, i] = s - (stv - P.h_min); G[:, i, IS] = 1; i += 1
    r[:, i] = s + P.tau_min * v - xs - stv; G[:, i, IS] = 1; G[:, i, IV] = P.tau_min; G[:, i, IXS] = -1; i += 1
    r[:, i] = s + v * T_HWP + v * v * G_HWP - xh - (stv - A_HWP); G[:, i, IS] = 1
    G[:, i, IV] = T_HWP + 2 * G_HWP * v; G[:, i, IXH] = -1
    curv.append((i, IV, IV, np.full(n, 2 * G_HWP))); i += 1
    r[:, i] = P.Fm_min - Fm; G[:, i, IFM] = -1; i += 1
    r[:, i] = Fm - P.Fm_max; G[:, i, IFM] = 1; i += 1
    r[:, i] = Fb; G[:, i, IFB] = 1; i += 1
    for q in (IXV, IXH, IXS, IXF):
        r[:, i] = -u[:, q - 4]; G[:, i, q] = -1; i += 1
    r[:, i] = -s; G[:, i, IS] = -1; i += 1
    r[:, i] = -v; G[:, i, IV] = -1; i += 1
    r[:, i] = v - V["v_max"]; G[:, i, IV] = 1; i += 1
    if math.isfinite(P.s_goal):
        r[:, i] = s - P.s_goal; G[:, i, IS] = 1; i += 1
    assert i == R
    return r, G, curv


def _stage_values(P: NlpProblem, chi, u, sigma):
    """Values only: scaled cost per stage, dynamics f(chi_k, u_k) [N][4], rows r [N][R]."""
    V = P.V
    N = P.N
    s, v, p, j = (chi[:-1, i] for i in range(4))
    Fm, F = u[:, 0], u[:, 0] + u[:, 1]
    th, _ = P.theta(s)
    s1, v1, q = P.rk4(s, v, Fm, F, np.cos(th), np.sin(th))
    th1, _ = P.theta(s1)
    p1 = (F - P.drag(v1, th1)) / (V["lambda"] * V["m"])
    f = np.stack([s1, v1, p1, (p1 - p) / P.Ts], axis=1)
    W = P.W
    cost = sigma * (q + P.Ts * (W[2] * j * j + W[3] * u[:, 2] + W[4] * (u[:, 3] ** 2 + 1e2 * u[:, 3])
                                + W[5] * u[:, 4] + W[6] * u[:, 5]))
    r, _, _ = _rows(P, chi[1:], u, np.arange(N))
    return cost, f, r


def _linearize(P: NlpProblem, chi, u, lam, nu, sigma):
    """Stage data of the Newton system.  Returns dict with cost, f, r, Jr, A, B, gl (cost gradient,
    10), Hl (Lagrangian Hessian without the barrier terms, in (chi_k, u_k)), Gc (row curvature in y)."""
    V, N, Ts, W = P.V, P.N, P.Ts, P.W
    lm = V["lambda"] * V["m"]
    s, v, p, j = (chi[:-1, i] for i in range(4))
    js, jv = Jet.var(s, 0), Jet.var(v, 1)
    jFm, jF = Jet.var(u[:, 0], 2), Jet.var(u[:, 0] + u[:, 1], 3)
    th, dth = P.theta(s)
    jth = Jet(th, dth * js.g, np.zeros((4, 4, N)))
    cth = jth.fn(np.cos(th), -np.sin(th), -np.cos(th))
    sth = jth.fn(np.sin(th), np.cos(th), -np.sin(th))
    s1, v1, q = P.rk4(js, jv, jFm, jF, cth, sth)
    th1, dth1 = P.theta(s1.v)
    jth1 = Jet(th1, dth1 * s1.g, dth1 * s1.h)
    c1 = jth1.fn(np.cos(th1), -np.sin(th1), -np.cos(th1))
    s1n = jth1.fn(np.sin(th1), np.cos(th1), -np.sin(th1))
    p1 = (jF - V["zeta_a"] * (v1 * v1) - (V["c_r"] * V["m"] * V["g"]) * c1 - (V["m"] * V["g"]) * s1n) * (1.0 / lm)
    f = np.stack([s1.v, v1.v, p1.v, (p1.v - p) / Ts], axis=1)
    gs, gv, gp = _jet_grad10(s1, N), _jet_grad10(v1, N), _jet_grad10(p1, N)
    AB = np.zeros((N, NX, NY))
    AB[:, 0], AB[:, 1], AB[:, 2] = gs, gv, gp
    AB[:, 3] = gp / Ts
    AB[:, 3, IP] -= 1.0 / Ts
    cost = sigma * (q.v + Ts * (W[2] * j * j + W[3] * u[:, 2] + W[4] * (u[:, 3] ** 2 + 1e2 * u[:, 3])
                                + W[5] * u[:, 4] + W[6] * u[:, 5]))
    gl = sigma * _jet_grad10(q, N)
    gl[:, IJ] += sigma * Ts * 2 * W[2] * j
    gl[:, IXV] += sigma * Ts * W[3]
    gl[:, IXH] += sigma * Ts * W[4] * (2 * u[:, 3] + 1e2)
    gl[:, IXS] += sigma * Ts * W[5]
    gl[:, IXF] += sigma * Ts * W[6]
    Hl = sigma * _jet_hess10(q, N)
    Hl[:, IJ, IJ] += sigma * Ts * 2 * W[2]
    Hl[:, IXH, IXH] += sigma * Ts * 2 * W[4]
    hp = _jet_hess10(p1, N)
    Hl += nu[1:, 0, None, None] * _jet_hess10(s1, N) + nu[1:, 1, None, None] * _jet_hess10(v1, N) \
        + (nu[1:, 2] + nu[1:, 3] / Ts)[:, None, None] * hp
    r, Jr, curv = _rows(P, chi[1:], u, np.arange(N))
    Gc = np.zeros((N, NY, NY))
    for (ri, a, b, val) in curv:
        Gc[:, a, b] += lam[:, ri] * val
        if a != b:
            Gc[:, b, a] += lam[:, ri] * val
    return dict(cost=cost, f=f, r=r, Jr=Jr, AB=AB, gl=gl, Hl=Hl, Gc=Gc)


REG_SCALE = np.array([1e-6, 1e-6, 1e-10, 1e-10, 1e-10, 1e-10])   # Levenberg term on the forces (per N^2); the slacks enter convexly


def _riccati(Q, q, AB, c, reg):
    """Backward / forward sweep.  Q [N][10][10], q [N][10], AB [N][4][10], c [N][4] (defects).
    Returns (ok, dchi [N+1][4], du [N][6], nu_new [N+1][4])."""
    N = Q.shape[0]
    P = np.zeros((NX, NX))
    p = np.zeros(NX)
    K = np.zeros((N, NU, NX))
    kf = np.zeros((N, NU))
    Ps = np.zeros((N + 1, NX, NX))
    ps = np.zeros((N + 1, NX))
    for k in range(N - 1, -1, -1):
        ab = AB[k]
        M = Q[k] + ab.T @ P @ ab
        M[np.arange(NX, NY), np.arange(NX, NY)] += reg * REG_SCALE
        m = q[k] + ab.T @ (P @ c[k] + p)
        Muu = M[NX:, NX:]
        L = np.zeros((NU, NU))
        for i in range(NU):                                   # Cholesky with a per-pivot relative test
            d = Muu[i, i] - L[i, :i] @ L[i, :i]
            if not d > 1e-10 * abs(Muu[i, i]):
                return False, None, None, None, None, None
            L[i, i] = math.sqrt(d)
            L[i + 1:, i] = (Muu[i + 1:, i] - L[i + 1:, :i] @ L[i, :i]) / L[i, i]
        sol = np.linalg.solve(L.T, np.linalg.solve(L, np.column_stack([M[NX:, :NX], m[NX:]])))
        K[k] = -sol[:, :NX]
        kf[k] = -sol[:, NX]
        P = M[:NX, :NX] + M[:NX, NX:] @ K[k]
        P = 0.5 * (P + P.T)
        p = m[:NX] + M[:NX, NX:] @ kf[k]
        Ps[k], ps[k] = P, p
    dchi = np.zeros((N + 1, NX))
    du = np.zeros((N, NU))
    nu = np.zeros((N + 1, NX))
    for k in range(N):
        du[k] = K[k] @ dchi[k] + kf[k]
        dchi[k + 1] = AB[k, :, :NX] @ dchi[k] + AB[k, :, NX:] @ du[k] + c[k]
        if k + 1 < N:
            nu[k + 1] = Ps[k + 1] @ dchi[k + 1] + ps[k + 1]
    return True, dchi, du, nu, K, kf


def _dyn_scalar(P: NlpProblem, s, v, p, F):
    """f(chi_k, u_k) for one stage in plain floats (dynamics only; the cost integral is not needed)."""
    V = P.V
    lm = V["lambda"] * V["m"]
    if P.T["flat"]:
        th = 0.0
    else:
        th = float(pwa(s, *P.T["slope"])[0])
    grav = V["c_r"] * V["m"] * V["g"] * math.cos(th) + V["m"] * V["g"] * math.sin(th)
    za = V["zeta_a"]
    DT = P.Ts / 4
    for _ in range(4):
        a1 = (F - za * v * v - grav) / lm
        v2 = v + (DT / 2) * a1
        a2 = (F - za * v2 * v2 - grav) / lm
        v3 = v + (DT / 2) * a2
        a3 = (F - za * v3 * v3 - grav) / lm
        v4 = v + DT * a3
        a4 = (F - za * v4 * v4 - grav) / lm
        s = s + (DT / 6) * (v + 2 * v2 + 2 * v3 + v4)
        v = v + (DT / 6) * (a1 + 2 * a2 + 2 * a3 + a4)
    th1 = 0.0 if P.T["flat"] else float(pwa(s, *P.T["slope"])[0])
    p1 = (F - za * v * v - V["c_r"] * V["m"] * V["g"] * math.cos(th1) - V["m"] * V["g"] * math.sin(th1)) / lm
    return s, v, p1, (p1 - p) / P.Ts


def rollout(P: NlpProblem, chi, u, chi_base, K, kf, alpha):
    """Closed-loop forward pass: u_k = u_base_k + alpha*kf_k + K_k (chi_k - chi_base_k), chi_{k+1} = f(chi_k, u_k)."""
    N = P.N
    cn = np.zeros((N + 1, NX))
    un = np.array(u, dtype=np.float64)
    cn[0] = chi[0]
    for k in range(N):
        if K is not None:
            un[k] = u[k] + alpha * kf[k] + K[k] @ (cn[k] - chi_base[k])
        cn[k + 1] = _dyn_scalar(P, cn[k, 0], cn[k, 1], cn[k, 2], un[k, 0] + un[k, 1])
    return cn, un


def assemble_newton(P: NlpProblem, chi, u, lam, t, nu, mu, sigma):
    """Stage data of the Newton system at (chi, u, lam, t): Q [N][10][10], q [N][10], AB [N][4][10], c [N][4] -- what
    `solve` hands to `_riccati` (and the GPU test hands to eepacc_nlp_riccati)."""
    N = P.N
    D = _linearize(P, chi, u, lam, nu, sigma)
    r, Jr, AB = D["r"], D["Jr"], D["AB"]
    c = D["f"] - chi[1:]
    rg = r + t
    Dg = lam / t
    T = np.zeros((N, NY, NY))
    T[:, np.arange(NX, NY), np.arange(NX, NY)] = 1.0
    T[:, :NX, :] = AB
    G = np.einsum("nri,nr,nrj->nij", Jr, Dg, Jr) + D["Gc"]
    gam = np.einsum("nri,nr->ni", Jr, mu / t + Dg * rg)
    cy = np.zeros((N, NY))
    cy[:, :NX] = c
    Q = D["Hl"] + np.einsum("nai,nab,nbj->nij", T, G, T)
    q = D["gl"] + np.einsum("nai,na->ni", T, gam + np.einsum("nab,nb->na", G, cy))
    return Q, q, AB, c


def initial_point(P: NlpProblem):
    """Starting trajectory: a plain car-following rollout (the reference starts IPOPT from z0 = 0, RunOpt_NLP.m:348;
    an interior-point method that keeps the dynamics satisfied needs a drivable start instead).  Speed target =
    min(speed limit - 1, desired-headway speed behind the lead vehicle), acceleration = (target - v)/2 s clipped
    to [-2, 1.2] m/s^2, force = lambda*m*a + resistance; slacks one unit above what the rows need."""
    N = P.N
    V = P.V
    lm = V["lambda"] * V["m"]
    chi = np.zeros((N + 1, NX))
    chi[0, 0], chi[0, 1] = P.s0, P.v0
    th0 = float(P.theta(np.array([P.s0]))[0][0])
    chi[0, 2] = -P.drag(P.v0, th0) / lm
    u = np.zeros((N, NU))
    for k in range(N):
        s, v, p = chi[k, 0], chi[k, 1], chi[k, 2]
        th = float(P.theta(np.array([s]))[0][0])
        vlim = float(pwa(s + 2.0 * v, *P.T["vlim"])[0])
        stop = float(pwa(s + 2.0 * v, *P.T["stop"])[0])
        gap = P.s_tv[min(k + 1, N - 1)] - A_HWP - 1.0 - s
        vh = max(0.0, gap / (T_HWP + 1.0))
        vt = max(0.0, min(vlim - 1.0, stop - 0.5, vh))
        a = min(1.2, max(-2.0, (vt - v) / 2.0))
        if v + a * P.Ts < 0.0:
            a = -v / P.Ts
        F = lm * a + float(P.drag(v, th))
        u[k, 0], u[k, 1] = (F, -1.0) if F > P.Fm_min * 0.5 else (P.Fm_min * 0.5, F - P.Fm_min * 0.5)
        u[k, 0] += 1.0
        chi[k + 1] = _dyn_scalar(P, s, v, p, u[k, 0] + u[k, 1])
    z = np.zeros((N, NU))
    z[:, :2] = u[:, :2]
    r0 = _rows(P, chi[1:], z, np.arange(N))[0]
    # slack needs of the rows (rows are  expr - xi <= 0): xi_f rows 0..11, xi_s 12 (+TL) and tau row, xi_v, xi_h
    nt = 2 * P.n_tl
    need_f = np.max(r0[:, 0:12], axis=1)
    need_s = np.maximum(np.max(r0[:, 12:13 + nt], axis=1), r0[:, 15 + nt])
    need_v = r0[:, 13 + nt]
    need_h = r0[:, 16 + nt]
    u[:, 2] = np.maximum(need_v, 0.0) + 1.0
    u[:, 3] = np.maximum(need_h, 0.0) + 1.0
    u[:, 4] = np.maximum(need_s, 0.0) + 1.0
    u[:, 5] = np.maximum(need_f, 0.0) + 1.0
    return chi, u


def solve(P: NlpProblem, opt: NlpOptions | None = None, start=None):
    """Returns dict(chi, u, J, iters, status, kkt=(dual, primal, compl), lam, nu, history)."""
    o = opt or NlpOptions()
    N, R, sigma = P.N, P.n_rows, o.obj_scale
    chi, u = start if start is not None else initial_point(P)
    chi, u = chi.copy(), u.copy()
    cost, f, r = _stage_values(P, chi, u, sigma)
    t = np.maximum(-r, 1e-2)
    mu = o.mu_init
    lam = mu / t
    nu = np.zeros((N + 1, NX))
    rho = 1.0
    reg_last = 0.0
    status = 1
    hist = []
    it = 0
    T = np.zeros((N, NY, NY))
    T[:, np.arange(NX, NY), np.arange(NX, NY)] = 1.0
    for it in range(o.max_iter):
        if o.primal:
            lam = np.where(r + t > 1e-9 * (1.0 + t), lam, mu / t)     # rows that hold: multiplier of the primal barrier
        D = _linearize(P, chi, u, lam, nu, sigma)
        r, Jr, AB = D["r"], D["Jr"], D["AB"]
        c = D["f"] - chi[1:]
        rg = r + t
        # residuals of the KKT system with the current multipliers
        gy = np.einsum("nri,nr->ni", Jr, lam)                     # rows, y coordinates (chi_{k+1}, u_k)
        # exact costates of the current point (adjoint recursion); the dual residual is the reduced gradient
        nu = np.zeros((N + 1, NX))
        for k in range(N - 1, -1, -1):
            nu[k + 1] = gy[k, :NX] + (D["gl"][k + 1, :NX] + AB[k + 1, :, :NX].T @ nu[k + 2] if k + 1 < N else 0.0)
        gu = D["gl"][:, NX:] + np.einsum("nxi,nx->ni", AB[:, :, NX:], nu[1:]) + gy[:, NX:]
        e_dual = float(np.abs(gu).max())
        e_prim = max(np.abs(c).max(), np.abs(rg).max())
        e_comp0 = np.abs(lam * t).max()
        e_compm = np.abs(lam * t - mu).max()
        hist.append((it, float(cost.sum() / sigma), e_dual, e_prim, e_comp0, mu))
        if o.verbose:
            print("it %3d J %.9e dual %.2e prim %.2e comp %.2e mu %.1e reg %.1e rho %.1e" %
                  (it, cost.sum() / sigma, e_dual, e_prim, e_comp0, mu, reg_last, rho))
        if max(e_dual, e_prim, e_comp0) <= o.tol:
            status = 0
            break
        while mu > o.mu_min and max(e_dual, e_prim, e_compm) <= o.kappa_eps * mu:
            mu = max(o.mu_min, min(o.kappa_mu * mu, mu ** o.theta_mu))
            e_compm = np.abs(lam * t - mu).max()
        # Newton system in stage form
        Dg = lam / t
        T[:, :NX, :] = AB
        G = np.einsum("nri,nr,nrj->nij", Jr, Dg, Jr) + D["Gc"]
        gam = np.einsum("nri,nr->ni", Jr, mu / t + Dg * rg)
        cy = np.zeros((N, NY))
        cy[:, :NX] = c
        Q = D["Hl"] + np.einsum("nai,nab,nbj->nij", T, G, T)
        q = D["gl"] + np.einsum("nai,na->ni", T, gam + np.einsum("nab,nb->na", G, cy))
        # chi_N carries nu[N] = 0 by construction (no stage after it): its stationarity is in stage N-1
        # Levenberg-Marquardt loop: the regularisation grows until the factorisation has the right inertia AND
        # the line search accepts a step of at least 1/8; it shrinks again after full steps
        reg = reg_last
        rows_i = rg > 1e-9 * (1.0 + t)
        infeas = float(np.sum(rg[rows_i]))
        accepted = False
        a = a_p = a_d = 0.0
        while True:
            ok, dchi, du, nu_new, K, kf = _riccati(Q, q, AB, c, reg)
            if ok:
                dy = np.concatenate([dchi[1:], du], axis=1)
                Jdy = np.einsum("nri,ni->nr", Jr, dy)
                dt = -rg - Jdy
                lam_new = mu / t + Dg * rg + Dg * Jdy
                dlam = lam_new - lam
                tau = max(o.tau_min, 1 - mu)
                neg = dt < 0
                a_p = min(1.0, float(np.min(np.where(neg, -tau * t / np.where(neg, dt, -1.0), np.inf))))
                neg = dlam < 0
                a_d = min(1.0, float(np.min(np.where(neg, -tau * lam / np.where(neg, dlam, -1.0), np.inf))))
                dphi_f = float(np.einsum("ni,ni->", D["gl"][:, NX:], du)
                               + np.einsum("ni,ni->", D["gl"][:, :NX], dchi[:-1]) - mu * np.sum(dt / t))
                if infeas > 0:
                    rho = max(rho, 1.1 * float(np.abs(lam_new[rows_i]).max()), dphi_f / (0.9 * infeas) + 1e-8)
                dphi = dphi_f - rho * infeas
                phi0 = cost.sum() - mu * np.sum(np.log(t)) + rho * infeas
                a = a_p
                for ls in range(o.max_ls):
                    chi_t, u_t = rollout(P, chi, u, chi, K, kf, a)
                    cost_t, f_t, r_t = _stage_values(P, chi_t, u_t, sigma)
                    t_t = np.where(rows_i, np.maximum(-r_t, t + a * dt), -r_t)
                    if np.all(t_t >= (1 - tau) * t):
                        inf_t = float(np.sum((r_t + t_t)[rows_i]))
                        phi_t = cost_t.sum() - mu * np.sum(np.log(t_t)) + rho * inf_t
                        if phi_t <= phi0 + 1e-4 * a * dphi + 1e-13 * abs(phi0):
                            accepted = True
                            break
                    a *= 0.5
                if accepted:
                    break
            reg = max(o.reg_first, reg * 8.0)
            if reg > o.reg_max:
                break
        if not accepted:
            status = 2
            break
        reg_last = reg / 3.0 if ls <= 1 else reg
        if reg_last < o.reg_first:
            reg_last = 0.0
        chi, u, t = chi_t, u_t, t_t
        cost = cost_t
        lam = lam + a_d * dlam
        lam = np.clip(lam, mu / (o.kappa_sigma * t), o.kappa_sigma * mu / t)
        if o.verbose:
            print("   a_p %.2e a_d %.2e alpha %.2e%s |du| %.2e |dchi| %.2e" %
                  (a_p, a_d, a, "" if accepted else " (not accepted)", np.abs(du).max(), np.abs(dchi).max()))
    W = P.W
    return dict(chi=chi, u=u, J=float(cost.sum() / sigma), iters=it, status=status, lam=lam, nu=nu,
                t=t, history=hist)
