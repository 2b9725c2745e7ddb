#!/usr/bin/env python3
"""Development prototype (numpy) of the structured FBMPC solver that csrc/eepacc_fbs_impl.inc implements.
Not product code, not the oracle: a readable model of the kernel's algorithm, used to debug the state
machine on the CPU and checked against the dense oracle by tests/test_proto_fb_structured.py.

Problem (ABO/Functions/MPCs/CreateQP_FB.m:158-489 condensed with the carried A(k)/D(k) of
ABO/RunOpt_FBMPC.m:247-259): per stage k the reference has the variables Fm_k, Fb_k and four slacks.
Here
    u_k = Fm_k + Fb_k   (total force: the only quantity the dynamics, the acceleration and jerk penalties and
                         most rows see; dense N x N coupling, positive definite)
    w_k = -Fb_k >= 0    (friction-brake share: no curvature of its own, price (c5 v_k + c2) that is bilinear
                         with the predicted speed -- this bilinear term is what makes the reference's dense
                         Hessian indefinite)
and every row is   n'(s_k, v_k, u_k, u_{k-1}) + aw*w_k - xi_group <= b.
The slacks are eliminated exactly as in the ABMPC solver (tools/proto_structured.py); w_k is treated the same
way: on its bound (w = 0) the rows that contain it are ordinary rows, off its bound it is defined by a pivot
row (torque limit, rear-axle limit or motor-force bound) and the bilinear term moves into the Hessian
(rank-2 update).  Keeping w pinned like this is the inertia control: the reduced Hessian stays positive
definite by construction of the working set.
"""
from __future__ import annotations

import math
import numpy as np

from proto_structured import estimate_traj, route_bounds, T_HWP, A_HWP, G_HWP


def interp_pwa(d, doms, vals):
    doms = np.asarray(doms, float); vals = np.asarray(vals, float)
    if d < doms[0]:
        return vals[0]
    if d > doms[-1]:
        return vals[-1]
    for i in range(len(doms) - 1):
        if doms[i] <= d <= doms[i + 1]:
            f = (d - doms[i]) / (doms[i + 1] - doms[i])
            return vals[i] + f * (vals[i + 1] - vals[i])
    return vals[-1]


class FBProblem:
    """u-space form of one FBMPC step.  A22, D2: carried arrays [N] (updated in place like the reference)."""

    def __init__(self, OPT, V, s0, v0, a_m1, t0, s_tv, v_tv, a_tv_prev, A22, D2, k_step, prev_sol=None):
        N = self.N = OPT["N_hor"]; Tvec = np.asarray(OPT["Tvec"], float)
        W = np.asarray(OPT["W_FB"], float)
        w_P, w_a, w_j, w_v, w_h, w_s, w_f = W
        self.w = dict(v=w_v, h=w_h, s=w_s, f=w_f)
        lm = V["lambda"] * V["m"]; za = V["zeta_a"]
        if OPT["paramEstSetting"] == 2:
            ps, pv = prev_sol
            se = np.concatenate([[s0], ps[2:], [ps[-1] + Tvec[-1] * pv[-1]]])
            ve = np.concatenate([[v0], pv[2:], [pv[-1]]])
        else:
            se, ve = estimate_traj(OPT, 0, s0, v0, a_m1)
        stv, _ = estimate_traj(OPT, 1, s_tv, v_tv, a_tv_prev)
        self.DistHor = se[N] - s0
        vlim, vcurv, vstop, vTL, amin, amax, jmin, jmax = route_bounds(OPT, se, ve, t0)
        theta = np.array([interp_pwa(se[k], OPT["s_slope"], OPT["slope"]) for k in range(N)])
        zrg = V["m"] * V["g"] * (V["c_r"] * np.cos(theta) + np.sin(theta))
        # A(k)/D(k) index quirk (RunOpt_FBMPC.m:247-259): MPC step k < N freezes stage k from v_est(N_hor)
        taylor = bool(OPT["FBuseTaylor"])
        if taylor:
            if k_step < N:
                i = N - 1
                A22[k_step] = 1.0 - 2.0 * Tvec[i] * za * ve[i] / lm
                D2[k_step] = Tvec[i] / lm * (za * ve[i] ** 2 - zrg[i])
        else:
            for i in range(N):
                D2[i] = Tvec[i] / lm * (-za * ve[i] ** 2 - zrg[i])
        beta = Tvec / lm
        # condensing: v_k = vbar_k + Sv[k] u ; s_k = sbar_k + Ss[k] u
        Sv = np.zeros((N + 1, N)); Ss = np.zeros((N + 1, N))
        vbar = np.zeros(N + 1); sbar = np.zeros(N + 1)
        vbar[0] = v0; sbar[0] = s0
        for k in range(N):
            Sv[k + 1] = A22[k] * Sv[k]; Sv[k + 1, k] += beta[k]
            Ss[k + 1] = Ss[k] + Tvec[k] * Sv[k]
            vbar[k + 1] = A22[k] * vbar[k] + D2[k]
            sbar[k + 1] = sbar[k] + Tvec[k] * vbar[k]
        self.Sv, self.Ss, self.vbar, self.sbar = Sv, Ss, vbar, sbar
        # sparse-form objective over y_k = (v_k, u_k) (CreateQP_FB.m:181-208), block tridiagonal
        K = (30.0 / math.pi) * V["phi"]
        bq = np.asarray(OPT["b_quadr"], float)
        assert bq[3] == 0.0, "b_quadr(4) != 0 gives w its own curvature: not covered by this solver"
        self.c5 = w_P * K * bq[4]; self.c2 = w_P * bq[1]
        Qvv = np.zeros((N, N)); Qvu = np.zeros((N, N)); Quu = np.zeros((N, N))   # Qvu[k,l]: v_k u_l
        cv = np.zeros(N); cu = np.zeros(N)
        for k in range(N):
            Qvv[k, k] += w_P * 2 * K * K * bq[5]; Qvu[k, k] += self.c5
            cv[k] += w_P * K * bq[2]; cu[k] += self.c2
            fa = 2 * w_a / lm ** 2
            Qvv[k, k] += fa * (za * za * ve[k] ** 2 + za * zrg[k]); Qvu[k, k] += fa * (-za * ve[k]); Quu[k, k] += fa
            cu[k] += w_a / lm ** 2 * (-2 * zrg[k])
            f = 2 * w_j / (lm * Tvec[k]) ** 2
            if k == 0:
                Quu[0, 0] += f
                cu[0] -= 2 * w_j * (za * v0 ** 2 + zrg[0] + lm * a_m1) / (lm * Tvec[0]) ** 2
            else:
                Dz = zrg[k] - zrg[k - 1]
                p, c = k - 1, k
                Qvv[p, p] += f * (za * za * ve[k] ** 2 + 2 * za * Dz)
                Qvu[p, p] += f * (-za * ve[k - 1])
                Qvv[p, c] += f * (-za * za * ve[k] * ve[k - 1]); Qvv[c, p] += f * (-za * za * ve[k] * ve[k - 1])
                Qvu[p, c] += f * (za * ve[k - 1])
                Quu[p, p] += f
                Qvu[c, p] += f * (za * ve[k])
                Quu[p, c] += -f; Quu[c, p] += -f
                Qvv[c, c] += f * (za * za * ve[k - 1] ** 2 - 2 * za * Dz)
                Qvu[c, c] += f * (-za * ve[k])
                Quu[c, c] += f
                cu[p] += f * Dz; cu[c] -= f * Dz
        S = Sv[:N]
        self.H = Quu + S.T @ Qvu + Qvu.T @ S + S.T @ Qvv @ S
        self.H = 0.5 * (self.H + self.H.T)
        self.g = cu + S.T @ cv + S.T @ (Qvv @ vbar[:N]) + Qvu.T @ vbar[:N]
        self.Q = (Qvv, Qvu, Quu, cv, cu)
        # rows
        rows = []
        def add(k, al, be, ga, de, aw, b, grp, name):
            rows.append(dict(k=k, al=al, be=be, ga=ga, de=de, aw=aw, b=b, grp=grp, name=name))
        c1 = V["phi"] * V["T_m_max"] ** 2 / 4 / V["P_m_max"]
        eta, phi = V["eta_TF"], V["phi"]
        Lmu, hg = V["L"] / V["mu"], V["h_g"]
        for k in range(N):
            T = Tvec[k]
            zw = V["m"] * V["g"] * (V["L_f"] * math.cos(theta[k]) + hg * math.sin(theta[k]))
            base = za * ve[k] ** 2 + zrg[k]
            add(k, -1, 0, 0, 0, 0, -s0, None, "s_lo")
            if math.isfinite(OPT["s_goal"]):
                add(k, 1, 0, 0, 0, 0, OPT["s_goal"], None, "s_hi")
            add(k, 0, -1, 0, 0, 0, -0.0, None, "v_lo")
            add(k, 0, 1, 0, 0, 0, V["v_max"], None, "v_hi")
            add(k, 0, 0, -1, 0, -1, 1e4, None, "fm_lo")
            add(k, 0, 0, 1, 0, 1, 1e4, None, "fm_hi")
            add(k, 0, 0, 0, 0, 1, 1e4, None, "fb_lo")
            add(k, 0, c1, -eta / phi, 0, -eta / phi, V["T_m_max"], ("f", k), "tq_min")
            add(k, 0, c1, 1 / eta / phi, 0, 1 / eta / phi, V["T_m_max"], ("f", k), "tq_max")
            add(k, 0, 0, -(Lmu + hg), 0, -Lmu, zw - hg * zrg[k], ("f", k), "rt_lo")
            add(k, 0, 0, Lmu - hg, 0, Lmu, zw - hg * zrg[k], ("f", k), "rt_hi")
            ft = V["mu"] * V["m"] * V["g"] * math.cos(theta[k])
            add(k, 0, 0, 1, 0, 0, ft, ("f", k), "ft_hi")
            add(k, 0, 0, -1, 0, 0, ft, ("f", k), "ft_lo")
            add(k, 0, 0, 1, 0, 0, lm * amax[k] + base, ("f", k), "amax")
            add(k, 0, 0, -1, 0, 0, -(lm * amin[k] + base), ("f", k), "amin")
            if k == 0:
                add(k, 0, 0, 1, 0, 0, lm * (T * jmax[k] + a_m1) + base, ("f", k), "jmax")
                add(k, 0, 0, -1, 0, 0, -(lm * (T * jmin[k] + a_m1) + base), ("f", k), "jmin")
            else:
                dj = za * (ve[k] ** 2 - ve[k - 1] ** 2) + (zrg[k] - zrg[k - 1])
                add(k, 0, 0, 1, -1, 0, lm * T * jmax[k] + dj, ("f", k), "jmax")
                add(k, 0, 0, -1, 1, 0, -(lm * T * jmin[k] + dj), ("f", k), "jmin")
            add(k, 0, 1, 0, 0, 0, vlim[k], ("f", k), "vlim")
            add(k, 0, 1, 0, 0, 0, vcurv[k], ("f", k), "vcurv")
            add(k, 0, 1, 0, 0, 0, vstop[k], ("s", k), "vstop")
            add(k, 0, 1, 0, 0, 0, vTL[k], ("s", k), "vTL")
            add(k, 0, -1, 0, 0, 0, -min(vlim[k], vcurv[k]), ("v", k), "vinc")
            add(k, 1, 0, 0, 0, 0, stv[k] - OPT["h_min"], ("s", k), "safe1")
            add(k, 1, OPT["tau_min"], 0, 0, 0, stv[k], ("s", k), "safe2")
            if taylor:
                add(k, 1, T_HWP + 2 * G_HWP * ve[k], 0, 0, 0, stv[k] - A_HWP + G_HWP * ve[k] ** 2, ("h", k), "hwp")
            else:
                add(k, 1, T_HWP + G_HWP * ve[k], 0, 0, 0, stv[k] - A_HWP, ("h", k), "hwp")
        add(N, 1, 0, 0, 0, 0, stv[N - 1] - OPT["h_min"], None, "term1")
        add(N, 1, OPT["tau_min"], 0, 0, 0, stv[N - 1], None, "term2")
        self.rows = []; self.groups = {}; self.infeasible_const = False
        for r in rows:
            k = r["k"]
            n = r["al"] * Ss[k] + r["be"] * Sv[k]
            if k < N:
                n = n.copy(); n[k] += r["ga"]
                if k > 0:
                    n[k - 1] += r["de"]
            ba = r["b"] - r["al"] * sbar[k] - r["be"] * vbar[k]
            grp = r["grp"]
            if grp is not None and grp not in self.groups:
                kind = grp[0]
                self.groups[grp] = dict(w=(100 * w_h if kind == "h" else self.w[kind]),
                                        q=(2 * w_h if kind == "h" else 0.0), lb=0.0, rows=[])
            if not np.any(n != 0) and r["aw"] == 0:      # constant row (stage 0 rows on s_0, v_0 only)
                if grp is None:
                    if -ba > 1e-9:
                        self.infeasible_const = True
                else:
                    self.groups[grp]["lb"] = max(self.groups[grp]["lb"], -ba)
                continue
            if not math.isfinite(ba):
                continue
            idx = len(self.rows)
            self.rows.append(dict(n=n, b=ba, grp=grp, aw=r["aw"], name=r["name"], k=k))
            if grp is not None:
                self.groups[grp]["rows"].append(idx)
        self.se, self.ve = se, ve

    # --- checks against the reference's dense formulation -----------------------------------------------
    def dense_x(self, u, w, xi):
        """x of the reference's dense QP (per stage Fm, Fb, xi_v, xi_h, xi_s, xi_f)"""
        N = self.N
        x = np.zeros(6 * N)
        x[0::6] = u + w; x[1::6] = -w
        for i, kind in enumerate("vhsf"):
            x[2 + i::6] = [xi[(kind, k)] for k in range(N)]
        return x

    def cost(self, u, w, xi):
        """value of the dense objective 1/2 x'Hx + g'x (sol.cost)"""
        c = 0.5 * u @ self.H @ u + self.g @ u
        v = self.vbar[:self.N] + self.Sv[:self.N] @ u
        c += np.sum((self.c5 * v + self.c2) * w)
        for grp, G in self.groups.items():
            x = xi[grp]
            c += G["w"] * x + 0.5 * G["q"] * x * x
        return c


RELAX = ("fm_lo", "tq_min", "rt_lo")        # rows that w relaxes (aw < 0): candidates for the pivot of w


class StructuredFB:
    """Dual active set in u-space with eliminated slacks and pinned brake shares."""

    def __init__(self, prob: FBProblem, tol=1e-9):
        self.p = prob
        self.N = prob.N
        self.tol = tol
        self.W = []
        self.lam = {}
        self.gstate = {g: dict(P=False, pivot=None, compl=False) for g in prob.groups}
        self.wst = [dict(P=False, pivot=None) for _ in range(prob.N)]
        self.iters = 0; self.events = 0
        self.refine_rounds = 2
        self.warm_tol = 1e-12
        self.single_passes = 8
        self.unsupported = False
        self.skipped_free = 0
        # rows per stage that contain w
        self.wrows = [[] for _ in range(prob.N)]
        for j, r in enumerate(prob.rows):
            if r["aw"] != 0:
                self.wrows[r["k"]].append(j)

    # ---- local-variable expressions (affine in u): value = n'u + c --------------------------------------
    def xi_expr(self, g):
        p = self.p; G = p.groups[g]; st = self.gstate[g]
        z = np.zeros(self.N)
        if G["q"] > 0:
            if st["compl"]:
                r = p.rows[G["rows"][0]]
                return r["n"], -r["b"]
            return z, G["lb"]
        if st["P"]:
            r = p.rows[st["pivot"]]
            if r["aw"] != 0 and self.wst[r["k"]]["P"]:
                # the pivot of xi_f contains w and w is off its bound: representable if w's own pivot is free of xi_f
                # (w first, then xi_f); both pivots among the coupled rows would need a 2 x 2 solve
                q = p.rows[self.wst[r["k"]]["pivot"]]
                if q["grp"] is not None:
                    self.unsupported = True
                    return r["n"], -r["b"]
                nw, cw = self.w_expr(r["k"])
                return r["n"] + r["aw"] * nw, -r["b"] + r["aw"] * cw
            return r["n"], -r["b"]
        return z, G["lb"]

    def w_expr(self, k):
        st = self.wst[k]
        if not st["P"]:
            return np.zeros(self.N), 0.0
        q = self.p.rows[st["pivot"]]
        a = -q["aw"]
        nx, cx = (self.xi_expr(q["grp"]) if q["grp"] is not None else (np.zeros(self.N), 0.0))
        return (q["n"] - nx) / a, (-q["b"] - cx) / a

    def eff(self, j):
        r = self.p.rows[j]
        n = r["n"].copy(); b = r["b"]
        if r["aw"] != 0:
            nw, cw = self.w_expr(r["k"])
            n = n + r["aw"] * nw; b = b - r["aw"] * cw
        if r["grp"] is not None:
            nx, cx = self.xi_expr(r["grp"])
            n = n - nx; b = b + cx
        return n, b

    def g_eff(self):
        p = self.p
        g = p.g.copy()
        for gk, st in self.gstate.items():
            G = p.groups[gk]
            if G["q"] == 0 and st["P"]:
                g += G["w"] * self.xi_expr(gk)[0]
            if G["q"] > 0 and st["compl"]:
                r = p.rows[G["rows"][0]]
                g += (G["w"] - G["q"] * r["b"]) * r["n"]
        for k in range(self.N):
            if self.wst[k]["P"]:
                nw, cw = self.w_expr(k)
                g += p.c5 * cw * p.Sv[k] + (p.c5 * p.vbar[k] + p.c2) * nw
        return g

    def H_eff(self):
        p = self.p
        H = p.H.copy()
        for gk, st in self.gstate.items():
            G = p.groups[gk]
            if G["q"] > 0 and st["compl"]:
                n = p.rows[G["rows"][0]]["n"]
                H += G["q"] * np.outer(n, n)
        for k in range(self.N):
            if self.wst[k]["P"]:
                nw, _ = self.w_expr(k)
                H += p.c5 * (np.outer(p.Sv[k], nw) + np.outer(nw, p.Sv[k]))
        return H

    def factor(self):
        He = self.H_eff()
        try:
            np.linalg.cholesky(He)
        except np.linalg.LinAlgError:
            self.unsupported = True
        self.Hinv = np.linalg.inv(He)
        m = len(self.W)
        if m == 0:
            self.C = np.zeros((0, self.N)); self.d = np.zeros(0); self.Pm = np.zeros((0, 0))
            return
        E = [self.eff(j) for j in self.W]
        self.C = np.array([e[0] for e in E]); self.d = np.array([e[1] for e in E])
        self.Pm = np.linalg.inv(self.C @ self.Hinv @ self.C.T)

    def primal(self, extra=None):
        rhs = self.g_eff()
        if len(self.W):
            rhs = rhs + self.C.T @ np.array([self.lam[j] for j in self.W])
        if extra is not None:
            rhs = rhs + extra[1] * extra[0]
        return -self.Hinv @ rhs

    def refine(self, extra=None, rounds=2):
        if not len(self.W):
            return
        for _ in range(rounds):
            a = self.primal(extra)
            res = self.C @ a - self.d
            dl = self.Pm @ res
            for j, x in zip(self.W, dl):
                self.lam[j] += x

    def solve_multipliers(self, extra=None):
        self.factor()
        if len(self.W):
            ge = self.g_eff()
            if extra is not None:
                ge = ge + extra[1] * extra[0]
            lam = -self.Pm @ (self.d + self.C @ (self.Hinv @ ge))
            for j, l in zip(self.W, lam):
                self.lam[j] = l

    # ---- values of the local variables at a point ---------------------------------------------------
    def xi(self, g, u):
        n, c = self.xi_expr(g)
        return n @ u + c

    def wval(self, k, u):
        n, c = self.w_expr(k)
        return n @ u + c

    # ---- multipliers of the pivots and of the bounds of the local variables -------------------------
    def local_mults(self, k, lamf, price, costs=True, q=None, lam_q=0.0):
        """stage k: lamf(j) multiplier (or rate) of working-set row j; price = cost coefficient of w_k (or its
        rate); costs: include the slack weights (False for rates).  q / lam_q: incoming constraint and its
        multiplier (or rate 1).  Returns dict: ('piv', group or 'w') -> pivot multiplier, ('mu', ...) ->
        bound multiplier of a local variable on its bound (only linear groups and w)."""
        p = self.p
        out = {}
        act = [j for j in self.W if p.rows[j]["k"] == k]
        def lq(cond):
            return lam_q if cond else 0.0
        q_row = q[1] if (q is not None and q[0] == "row" and p.rows[q[1]]["k"] == k) else None
        wst = self.wst[k]
        # w first when its pivot does not depend on a P-state xi_f through coupled rows (guaranteed)
        sw = price + sum(p.rows[j]["aw"] * lamf(j) for j in act if p.rows[j]["aw"] != 0)
        if q_row is not None and p.rows[q_row]["aw"] != 0:
            sw += p.rows[q_row]["aw"] * lam_q
        if q is not None and q[0] == "wbound" and q[1] == k:
            sw -= lam_q
        gF = ("f", k)
        stF = self.gstate[gF]
        lam_pF = None
        if stF["P"] and p.rows[stF["pivot"]]["aw"] != 0:
            # xi_f pivot contains w (then w is on its bound): pivot multiplier first, it feeds into mu_w
            sF = (p.groups[gF]["w"] if costs else 0.0) - sum(lamf(j) for j in act if p.rows[j]["grp"] == gF)
            sF -= lq(q_row is not None and p.rows[q_row]["grp"] == gF)
            sF -= lq(q is not None and q[0] == "bound" and q[1] == gF)
            lam_pF = sF
            sw += p.rows[stF["pivot"]]["aw"] * lam_pF
        lam_pW = None
        if wst["P"]:
            a = -p.rows[wst["pivot"]]["aw"]
            lam_pW = sw / a
            out[("piv", "w")] = lam_pW
        else:
            out[("mu", "w")] = sw
        for kind in "fsv":
            g = (kind, k)
            if g not in p.groups:
                continue
            st = self.gstate[g]
            if g == gF and lam_pF is not None:
                out[("piv", g)] = lam_pF
                continue
            s = (p.groups[g]["w"] if costs else 0.0) - sum(lamf(j) for j in act if p.rows[j]["grp"] == g)
            s -= lq(q_row is not None and p.rows[q_row]["grp"] == g)
            s -= lq(q is not None and q[0] == "bound" and q[1] == g)
            if g == gF and wst["P"] and p.rows[wst["pivot"]]["grp"] == gF:
                s -= lam_pW
            out[("piv", g) if st["P"] else ("mu", g)] = s
        return out

    def w_relevant(self, k, q=None):
        if self.wst[k]["P"]:
            return True
        if any(self.p.rows[j]["k"] == k and self.p.rows[j]["aw"] != 0 for j in self.W):
            return True
        stF = self.gstate[("f", k)]
        if stF["P"] and self.p.rows[stF["pivot"]]["aw"] != 0:
            return True
        if q is not None and q[0] == "row" and self.p.rows[q[1]]["k"] == k and self.p.rows[q[1]]["aw"] != 0:
            return True
        return False

    # ---- constraint selection ---------------------------------------------------------------------------
    def most_violated(self, u):
        p = self.p
        best, bq = self.tol, None
        inW = set(self.W)
        v = p.vbar[:self.N] + p.Sv[:self.N] @ u
        for j, r in enumerate(p.rows):
            if j in inW:
                continue
            g = r["grp"]
            if g is not None and self.gstate[g]["P"] and self.gstate[g]["pivot"] == j:
                continue
            if g is not None and p.groups[g]["q"] > 0 and self.gstate[g]["compl"]:
                continue
            k = r["k"]
            if r["aw"] != 0 and self.wst[k]["P"] and self.wst[k]["pivot"] == j:
                continue
            # w is free while its price is not positive (predicted speed below zero: a transient of the dual
            # iteration, v_k >= 0 is a hard row): rows that w relaxes cost nothing to satisfy then
            if r["aw"] < 0 and p.c5 * v[k] + p.c2 <= 0.0:
                self.skipped_free += 1
                continue
            val = r["n"] @ u - r["b"]
            if r["aw"] != 0:
                val += r["aw"] * self.wval(k, u)
            if g is not None:
                val -= self.xi(g, u)
            sc = val / (1.0 + abs(r["b"]))
            if sc > best:
                best, bq = sc, ("row", j)
        for g, st in self.gstate.items():
            G = p.groups[g]
            if st["P"] or (G["q"] > 0 and st["compl"]):
                val = G["lb"] - self.xi(g, u)
                if val > best:
                    best, bq = val, ("bound", g)
        for k in range(self.N):
            if self.wst[k]["P"]:
                val = -self.wval(k, u) / 1e3        # force units: scaled like a row with |b| ~ 1e3
                if val > best:
                    best, bq = val, ("wbound", k)
        return bq

    def incoming_eff(self, q):
        p = self.p
        if q[0] == "row":
            return self.eff(q[1])
        if q[0] == "wbound":
            n, c = self.w_expr(q[1])
            return -n, c                      # -w <= 0  <=>  -n'u <= c
        g = q[1]
        n, c = self.xi_expr(g)
        return -n, c - p.groups[g]["lb"]       # lb - xi <= 0

    # ---- dual feasibility of a given state (warm start, or after the one discontinuous event "cap_in") -------
    def reset_cold(self):
        self.W = []; self.lam = {}
        for st in self.gstate.values():
            st.update(P=False, pivot=None, compl=False)
        for st in self.wst:
            st.update(P=False, pivot=None)

    def repair(self, max_pass=40, verbose=False):
        """multipliers of the current state; while one of them has the wrong sign fix the worst offender (drop the
        row / switch a quadratic slack to its penalty / move a pivot) and recompute.  Returns the primal point."""
        p = self.p
        for it in range(max_pass):
            try:
                self.solve_multipliers()
            except np.linalg.LinAlgError:
                self.reset_cold(); continue
            if self.refine_rounds:
                self.refine(None, self.refine_rounds)
            u = self.primal()
            if self.unsupported:
                return u
            lam_max = max([abs(v) for v in self.lam.values()] + [0.0])
            tol = self.warm_tol * (1.0 + lam_max)
            worst, fix = tol, None
            for j in self.W:
                g = p.rows[j]["grp"]; lam = self.lam[j]
                if -lam > worst:
                    worst, fix = -lam, ("drop", j)
                if g is not None and p.groups[g]["q"] > 0 and lam - p.groups[g]["w"] > worst:
                    worst, fix = lam - p.groups[g]["w"], ("compl", j)
            v = p.vbar[:self.N] + p.Sv[:self.N] @ u
            for k in range(self.N):
                wrel = self.w_relevant(k)
                val = self.local_mults(k, lambda j: self.lam[j], p.c5 * v[k] + p.c2, True)
                for key, x in val.items():
                    if key[1] == "w" and not wrel:
                        continue
                    scale = 1.0 if key[1] == "w" else (1.0 + p.groups[key[1]]["w"])
                    if -x > worst * scale:
                        worst, fix = -x / scale, ("local", k, key)
            if fix is None:
                return u
            if verbose:
                print("  repair", it, fix, worst)
            if fix[0] == "drop":
                self.W.remove(fix[1]); del self.lam[fix[1]]
            elif fix[0] == "compl":
                self.W.remove(fix[1]); del self.lam[fix[1]]
                self.gstate[p.rows[fix[1]]["grp"]]["compl"] = True
            else:
                k, key = fix[1], fix[2]
                var = key[1]; is_w = var == "w"
                st = self.wst[k] if is_w else self.gstate[var]
                cands = self.members(k, var, ("none",), 0.0)
                if not is_w and self.wst[k]["P"] and p.rows[self.wst[k]["pivot"]]["grp"] is not None:
                    cands = [c for c in cands if p.rows[c[0]]["aw"] == 0] or cands
                if cands:
                    j = max(cands, key=lambda c: c[1])[0]
                    self.W.remove(j); del self.lam[j]
                    st["P"] = True; st["pivot"] = j
                else:
                    st["P"] = False; st["pivot"] = None
        self.reset_cold()
        self.solve_multipliers()
        return self.primal()

    # ---- main loop ----------------------------------------------------------------------------------------
    def solve(self, max_iter=2000, verbose=False):
        p = self.p
        self.status = 0
        self.factor()
        while True:
            u = self.repair(verbose=verbose)
            q = self.most_violated(u)
            if q is None or self.unsupported:
                break
            self.iters += 1
            if self.iters > max_iter:
                self.status = 2; break
            lam_q = 0.0
            done = False
            while not done:
                self.events += 1
                if self.events > 20 * max_iter or self.unsupported:
                    self.status = 2; done = True; break
                c, dq = self.incoming_eff(q)
                self.solve_multipliers((c, lam_q))
                if self.refine_rounds:
                    self.refine((c, lam_q), self.refine_rounds)
                u = self.primal((c, lam_q))
                viol = c @ u - dq
                uu = self.Hinv @ c
                m = len(self.W)
                s = self.C @ uu if m else np.zeros(0)
                r = self.Pm @ s if m else np.zeros(0)
                zz = c @ uu - (s @ r if m else 0.0)
                t2 = viol / zz if zz > 1e-8 * (c @ uu) else math.inf
                if viol <= 0:
                    t2 = 0.0
                # primal direction per unit of the incoming multiplier: u(t) = u - t z
                z = uu - (self.Hinv @ (self.C.T @ r) if m else 0.0)
                vrate = -(p.Sv[:self.N] @ z)
                v = p.vbar[:self.N] + p.Sv[:self.N] @ u
                t1, ev = math.inf, None
                rate_of = {j: -r[i] for i, j in enumerate(self.W)}
                for i, j in enumerate(self.W):
                    rj = p.rows[j]; g = rj["grp"]
                    lam = self.lam[j]
                    if g is not None and p.groups[g]["q"] > 0:
                        G = p.groups[g]
                        if r[i] > 0:
                            t = max(lam, 0.0) / r[i]
                            if t < t1: t1, ev = t, ("drop", j)
                        elif r[i] < 0:
                            t = max(G["w"] - lam, 0.0) / (-r[i])
                            if t < t1: t1, ev = t, ("compl", j)
                    elif r[i] > 0:
                        t = max(lam, 0.0) / r[i]
                        if t < t1: t1, ev = t, ("drop", j)
                if q[0] == "bound" and p.groups[q[1]]["q"] > 0:
                    G = p.groups[q[1]]
                    xi_now = G["lb"] - viol
                    den = 1.0 - G["q"] * zz
                    if den > 0:
                        t = max(G["w"] + G["q"] * xi_now - lam_q, 0.0) / den
                        if t < t1: t1, ev = t, ("drop_h", q[1])
                if q[0] == "row" and p.rows[q[1]]["grp"] is not None and p.groups[p.rows[q[1]]["grp"]]["q"] > 0 \
                        and not self.gstate[p.rows[q[1]]["grp"]]["compl"]:
                    t = max(p.groups[p.rows[q[1]]["grp"]]["w"] - lam_q, 0.0)
                    if t < t1: t1, ev = t, ("cap_in", p.rows[q[1]]["grp"])
                # local variables: bound multipliers (on the bound) / pivot multipliers (off it) must stay >= 0
                for k in range(self.N):
                    wrel = self.w_relevant(k, q)
                    price = p.c5 * v[k] + p.c2
                    val = self.local_mults(k, lambda j: self.lam[j], price, True, q, lam_q)
                    rat = self.local_mults(k, lambda j: rate_of[j], p.c5 * vrate[k], False, q, 1.0)
                    for key, x in val.items():
                        if key[1] == "w" and not wrel:
                            continue
                        dx = rat[key]
                        if dx < 0:
                            t = max(x, 0.0) / (-dx)
                            if t < t1: t1, ev = t, ("local", k, key)
                t = min(t1, t2)
                if not math.isfinite(t):
                    self.status = 1; done = True; break
                for i, j in enumerate(self.W):
                    self.lam[j] -= t * r[i]
                lam_q += t
                if verbose:
                    print("  it", self.iters, "q", q, (p.rows[q[1]]["name"], p.rows[q[1]]["k"]) if q[0] == "row" else "",
                          "viol %.3e t %.3e" % (viol, t), "ev", ev if t1 <= t2 else "full", "m", m)
                if t2 <= t1:
                    self.add_incoming(q, lam_q)
                    done = True
                else:
                    done = self.handle_event(ev, q, lam_q)
        self.solve_multipliers()
        self.refine(None, 3)
        self.u = self.primal()
        if self.unsupported:
            self.status = 4
        return self.status

    # ------------------------------------------------------------------------------
    def add_incoming(self, q, lam_q):
        p = self.p
        if q[0] == "row":
            j = q[1]
            r = p.rows[j]
            if r["name"] == "fb_lo" and self.wst[r["k"]]["P"]:
                # w reaches its upper bound: the bound row becomes w's pivot (canonical form of "w on its upper bound",
                # free of xi_f, so that rows containing w may define xi_f), the former pivot turns into an ordinary row
                st = self.wst[r["k"]]
                old = st["pivot"]
                st["pivot"] = j
                self.W.append(old); self.lam[old] = 0.0
                return
            self.W.append(j); self.lam[j] = lam_q
            return
        if q[0] == "wbound":
            st = self.wst[q[1]]
            piv = st["pivot"]
            st["P"] = False; st["pivot"] = None
            self.W.append(piv); self.lam[piv] = 0.0
            return
        g = q[1]
        if p.groups[g]["q"] > 0:
            self.gstate[g]["compl"] = False
            j = p.groups[g]["rows"][0]
            self.W.append(j); self.lam[j] = max(p.groups[g]["w"] - lam_q, 0.0)
            return
        st = self.gstate[g]
        piv = st["pivot"]
        st["P"] = False; st["pivot"] = None
        self.W.append(piv); self.lam[piv] = 0.0

    def members(self, k, key, q, lam_q):
        """candidate takeover rows of local variable `key` at stage k: (row, weighted multiplier, is_incoming)"""
        p = self.p
        out = []
        for j in self.W:
            r = p.rows[j]
            if r["k"] != k:
                continue
            if key == "w":
                if r["aw"] < 0:
                    out.append((j, -r["aw"] * self.lam[j], False))
            elif r["grp"] == key:
                out.append((j, self.lam[j], False))
        if q[0] == "row" and p.rows[q[1]]["k"] == k:
            r = p.rows[q[1]]
            if key == "w" and r["aw"] < 0:
                out.append((q[1], -r["aw"] * lam_q, True))
            elif key != "w" and r["grp"] == key:
                out.append((q[1], lam_q, True))
        return out

    def handle_event(self, ev, q, lam_q):
        p = self.p
        kind = ev[0]
        if kind == "drop":
            j = ev[1]
            self.W.remove(j); del self.lam[j]
            return False
        if kind == "compl":
            j = ev[1]; g = p.rows[j]["grp"]
            self.W.remove(j); del self.lam[j]
            self.gstate[g]["compl"] = True
            return False
        if kind == "drop_h":
            self.gstate[ev[1]]["compl"] = False
            return True
        if kind == "cap_in":
            self.gstate[ev[1]]["compl"] = True
            return True
        if kind == "local":
            k, key = ev[1], ev[2]
            var = key[1]                      # 'w' or a group
            is_w = var == "w"
            st = self.wst[k] if is_w else self.gstate[var]
            cands = self.members(k, var, q, lam_q)
            if not is_w and self.wst[k]["P"] and p.rows[self.wst[k]["pivot"]]["grp"] is not None:
                # a pivot of xi_f that contains w while w's pivot contains xi_f would couple the two eliminations
                cands = [c for c in cands if p.rows[c[0]]["aw"] == 0] or cands
            if is_w:
                stF = self.gstate[("f", k)]
                if stF["P"] and p.rows[stF["pivot"]]["aw"] != 0:
                    # xi_f is defined through a row that contains w: w may only be pinned by a row free of xi_f
                    cands = [c for c in cands if p.rows[c[0]]["grp"] is None]
                    if not cands:
                        self.unsupported = True
            q_bound_here = (q[0] == "wbound" and is_w and q[1] == k) or (q[0] == "bound" and q[1] == var)
            if key[0] == "mu":                 # bound multiplier reached zero: the variable leaves its bound
                if not cands:
                    # nothing defines the variable: only possible for w when its price vanished
                    return False if is_w else True
                j, _, inc = max(cands, key=lambda c: c[1])
                st["P"] = True; st["pivot"] = j
                if inc:
                    return True
                self.W.remove(j); del self.lam[j]
                return False
            # pivot multiplier reached zero
            old = st["pivot"]
            if cands:
                j, _, inc = max(cands, key=lambda c: c[1])
                st["pivot"] = j
                if inc:
                    return True
                self.W.remove(j); del self.lam[j]
                return False
            if q_bound_here:
                st["P"] = False; st["pivot"] = None
                return True
            # price of w vanished (v_k fell to zero along the path): w is free, the pivot row is let go
            st["P"] = False; st["pivot"] = None
            return False
        raise RuntimeError(ev)

    # ------------------------------------------------------------------------------
    def solution(self):
        u = self.u
        xi = {g: max(self.xi(g, u), self.p.groups[g]["lb"]) for g in self.p.groups}
        w = np.array([max(self.wval(k, u), 0.0) for k in range(self.N)])
        return u, w, xi


def solve_fb_step(OPT, V, s0, v0, a_m1, t0, s_tv, v_tv, a_tv_prev, A22, D2, k_step, prev_sol=None, verbose=False):
    prob = FBProblem(OPT, V, s0, v0, a_m1, t0, s_tv, v_tv, a_tv_prev, A22, D2, k_step, prev_sol)
    qp = StructuredFB(prob)
    st = qp.solve(verbose=verbose)
    return prob, qp, st
