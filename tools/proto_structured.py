#!/usr/bin/env python3
"""Development prototype (numpy) of the structured dual active-set solver that the HIP kernel
implements.  Not product code, not the oracle: a readable model of the kernel's algorithm used
to debug the state machine on the CPU before writing HIP.  See DESIGN.md section "QP solve".

Problem (AB): variables a in R^N; every constraint row of the reference QP
(ABO/Functions/MPCs/CreateQP_AB.m:256-387) has the form
      al*s_k + be*v_k + ga*a_k + de*a_{k-1} - xi_g <= b
with s_k, v_k affine in a (double integrator), xi_g the slack of the row's group g (or no slack
for hard rows).  Slacks are eliminated: linear-cost groups become capped-multiplier groups,
the quadratic-cost slack xi_h becomes a compliant row.
"""
from __future__ import annotations

import math
import numpy as np

T_HWP, A_HWP = 2.0, 2.0
G_HWP = -0.0246 * T_HWP + 0.010819


# ----------------------------------------------------------------------------------------
def estimate_traj(OPT, est, s, v, a):
    N = OPT["N_hor"]; Tvec = OPT["Tvec"]
    mode = OPT["paramEstSetting"] if est == 0 else OPT["TVestSetting"]
    tc = OPT["tConstACC_tar"] if est == 1 else OPT["tConstACC_ego"]
    se = np.zeros(N + 1); ve = np.zeros(N + 1)
    if mode == 0:
        se[0] = s
        for i in range(1, N + 1):
            se[i] = se[i - 1] + Tvec[i - 1] * v
        ve[:] = v
    elif mode == 1:
        se[0] = s; ve[0] = v
        for i in range(1, N + 1):
            Ts = Tvec[i - 1]
            if (i + 1) <= tc / Ts and ve[i - 1] + Ts * a > 0:
                ve[i] = ve[i - 1] + Ts * a
            else:
                ve[i] = ve[i - 1]
            se[i] = se[i - 1] + Ts * ve[i - 1]
    else:
        raise NotImplementedError
    return se, ve


def route_bounds(OPT, se, ve, t0):
    N = OPT["N_hor"]
    ssl, vsl = OPT["s_speedLim"], OPT["v_speedLim"]
    sc, cu = OPT["s_curv"], OPT["curvature"]
    vlim = np.zeros(N); vcurv = np.zeros(N); vstop = np.full(N, 1e5); vTL = np.full(N, 1e5)
    amin = np.zeros(N); amax = np.zeros(N); jmin = np.zeros(N); jmax = np.zeros(N)
    for i in range(N):
        for j in range(len(ssl)):
            if j == len(ssl) - 1:
                vlim[i] = ssl[-1]
            elif ssl[j] <= se[i] < ssl[j + 1]:
                vlim[i] = vsl[j]; break
        for j in range(len(sc)):
            if j == len(sc) - 1:
                vcurv[i] = OPT["alpha_TTL"] * abs(cu[-1]) ** (-1 / 3)
            elif sc[j] < se[i] < sc[j + 1]:
                vcurv[i] = OPT["alpha_TTL"] * abs(cu[j]) ** (-1 / 3); break
        for sl in np.atleast_1d(OPT.get("stopLoc", [])):
            d = abs(sl - se[i])
            if d < OPT["stopRefDist"]:
                vstop[i] = d * OPT["stopRefVelSlope"] + OPT["stopVel"]
        for TL in np.asarray(OPT.get("TLLoc", np.zeros((0, 4)))).reshape(-1, 4):
            x = t0 + (i + 1) * OPT["Tvec"][i] - TL[1]
            m = TL[2] + TL[3]
            if x - math.floor(x / m) * m < TL[2]:
                d = TL[0] - se[i]
                if abs(d) < OPT["stopRefDist"]:
                    if d < 0:
                        vTL[i] = abs(d) * OPT["stopRefVelSlope"] + OPT["TLstopVel"]
                    elif abs(d) < OPT["TLStopRegionSize"]:
                        vTL[i] = OPT["TLstopVel"]
                    else:
                        vTL[i] = abs(d - OPT["stopVel"]) * OPT["stopRefVelSlope"] + OPT["TLstopVel"]
        if ve[i] < 5:
            amin[i], amax[i], jmin[i], jmax[i] = -5, 4, -5, 5
        elif ve[i] < 20:
            amin[i] = -5.5 + ve[i] / 10; amax[i] = 14 / 3 - 2 * ve[i] / 15
            jmin[i] = -35 / 6 + ve[i] / 6; jmax[i] = 35 / 6 - ve[i] / 6
        else:
            amin[i], amax[i], jmin[i], jmax[i] = -3.5, 2, -2.5, 2.5
    return vlim, vcurv, vstop, vTL, amin, amax, jmin, jmax


# ----------------------------------------------------------------------------------------
class ABProblem:
    """a-space form of one ABMPC step."""

    def __init__(self, OPT, V, s0, v0, a_m1, t0, s_tv, v_tv, a_tv_prev):
        N = self.N = OPT["N_hor"]; Tvec = np.asarray(OPT["Tvec"], float)
        W = np.asarray(OPT["W_AB"], float)
        if W.size == 6:
            W = np.concatenate([[0.0], W])
        w_FC, w_a, w_j, w_v, w_h, w_s, w_f = W
        self.w = dict(v=w_v, h=w_h, s=w_s, f=w_f)
        se, ve = estimate_traj(OPT, 0, s0, v0, a_m1)
        stv, _ = estimate_traj(OPT, 1, s_tv, v_tv, a_tv_prev)
        self.DistHor = se[N] - s0
        vlim, vcurv, vstop, vTL, amin, amax, jmin, jmax = route_bounds(OPT, se, ve, t0)
        # condensing matrices (N+1 x N): v_k = v0 + Sv[k]@a, s_k = s0 + tau_k v0 + Ss[k]@a
        Sv = np.zeros((N + 1, N)); Ss = np.zeros((N + 1, N))
        for k in range(N):
            Sv[k + 1] = Sv[k]; Sv[k + 1, k] += Tvec[k]
            Ss[k + 1] = Ss[k] + Tvec[k] * Sv[k]; Ss[k + 1, k] += 0.5 * Tvec[k] ** 2
        tau = np.concatenate([[0], np.cumsum(Tvec)])
        self.Sv, self.Ss = Sv, Ss
        self.sf = s0 + tau * v0        # free response
        self.vf = np.full(N + 1, v0)
        # objective
        cq = w_FC * V["p01"] * V["F2"] if W[0] != 0 else 0.0
        H = 2 * cq * Sv[:N].T @ Sv[:N] + 2 * w_a * np.eye(N)
        g = Sv[:N].T @ (2 * cq * self.vf[:N] + w_FC * V["p10"]) + w_FC * V["p01"] * V["lambda"] * V["m"]
        for k in range(N):
            qj = 2 * w_j / Tvec[k] ** 2
            H[k, k] += qj
            if k > 0:
                H[k - 1, k - 1] += qj; H[k, k - 1] -= qj; H[k - 1, k] -= qj
        g[0] -= 2 * w_j / Tvec[0] * a_m1
        self.H, self.g = H, g
        self.Hinv = np.linalg.inv(H)
        # rows: (k, al, be, ga, de, b, group)   group = (kind, k) or None
        rows = []
        def add(k, al, be, ga, de, b, grp, name):
            rows.append(dict(k=k, al=al, be=be, ga=ga, de=de, b=b, grp=grp, name=name))
        inf = math.inf
        route_rows = bool(OPT.get("ab_route_rows", OPT.get("tree") == "ORIG"))
        for k in range(N):
            T = Tvec[k]
            add(k, -1, 0, 0, 0, -0.0, None, "s_lo")
            if math.isfinite(OPT["s_goal"]):
                add(k, 1, 0, 0, 0, OPT["s_goal"], None, "s_hi")
            add(k, 0, -1, 0, 0, -0.0, None, "v_lo")
            add(k, 0, 1, 0, 0, V["v_max"], None, "v_hi")
            add(k, 0, 0, 1, 0, amax[k], ("f", k), "amax")
            add(k, 0, 0, -1, 0, -amin[k], ("f", k), "amin")
            if k > 0:
                add(k, 0, 0, 1, -1, T * jmax[k], ("f", k), "jmax")
                add(k, 0, 0, -1, 1, -T * jmin[k], ("f", k), "jmin")
            else:
                add(k, 0, 0, 1, 0, T * jmax[k] + a_m1, ("f", k), "jmax")
                add(k, 0, 0, -1, 0, -(T * jmin[k] + a_m1), ("f", k), "jmin")
            if route_rows:
                add(k, 0, 1, 0, 0, vlim[k], ("f", k), "vlim")
                add(k, 0, 1, 0, 0, vcurv[k], ("f", k), "vcurv")
                add(k, 0, 1, 0, 0, vstop[k], ("s", k), "vstop")
                add(k, 0, 1, 0, 0, vTL[k], ("s", k), "vTL")
            add(k, 0, -1, 0, 0, -min(vlim[k], vcurv[k]), ("v", k), "vinc")
            add(k, 1, 0, 0, 0, stv[k] - OPT["h_min"], ("s", k), "safe1")
            add(k, 1, OPT["tau_min"], 0, 0, stv[k], ("s", k), "safe2")
            add(k, 1, T_HWP + G_HWP * ve[k], 0, 0, stv[k] - A_HWP, ("h", k), "hwp")
        add(N, 1, 0, 0, 0, stv[N - 1] - OPT["h_min"], None, "term1")
        add(N, 1, OPT["tau_min"], 0, 0, stv[N - 1], None, "term2")
        # a-space normals and rhs
        self.rows = []
        self.groups = {}
        self.infeasible_const = False
        for r in rows:
            k = r["k"]
            n = r["al"] * Ss[k] + r["be"] * Sv[k]
            if k < N:
                n = n.copy(); n[k] += r["ga"]
                if k > 0:
                    n[k - 1] += r["de"]
            ba = r["b"] - r["al"] * self.sf[k] - r["be"] * self.vf[k]
            grp = r["grp"]
            if grp is not None and grp not in self.groups:
                kind = grp[0]
                self.groups[grp] = dict(w=(100 * w_h if kind == "h" else self.w[kind]),
                                        q=(2 * w_h if kind == "h" else 0.0), lb=0.0, rows=[])
            if not np.any(n != 0):          # constant row
                if grp is None:
                    if -ba > 1e-9:
                        self.infeasible_const = True
                else:
                    self.groups[grp]["lb"] = max(self.groups[grp]["lb"], -ba)
                continue
            if not math.isfinite(ba):
                continue
            idx = len(self.rows)
            self.rows.append(dict(n=n, b=ba, grp=grp, name=r["name"], k=k))
            if grp is not None:
                self.groups[grp]["rows"].append(idx)

    # objective value of the reference's dense QP (sol.cost): f(a, xi) - f(0, 0)
    def dense_cost(self, a, xi):
        c = 0.5 * a @ self.H @ a + self.g @ a
        for grp, G in self.groups.items():
            x = xi[grp]
            c += G["w"] * x + 0.5 * G["q"] * x * x
        return c


# ----------------------------------------------------------------------------------------
class StructuredQP:
    """Dual active set with capped-multiplier groups (see module docstring)."""

    def __init__(self, prob: ABProblem, tol=1e-9):
        self.p = prob
        self.N = prob.N
        self.tol = tol
        self.Hinv = prob.Hinv
        self.W = []            # list of row indices in the working set (non-pivot)
        self.lam = {}          # row -> multiplier (rows in W)
        self.gstate = {g: dict(P=False, pivot=None, compl=False) for g in prob.groups}
        self.iters = 0
        self.events = 0
        self.refine_rounds = 2
        self.warm_tol = 1e-12
        self.single_passes = 8

    # effective (a-space) normal / rhs / compliance of a working-set row
    def eff(self, j):
        r = self.p.rows[j]
        g = r["grp"]
        if g is None:
            return r["n"], r["b"], 0.0
        G = self.p.groups[g]; st = self.gstate[g]
        if G["q"] > 0:                      # quadratic slack: in the working set only while xi = lb
            return r["n"], r["b"] + G["lb"], 0.0
        if st["P"]:
            pr = self.p.rows[st["pivot"]]
            return r["n"] - pr["n"], r["b"] - pr["b"], 0.0
        return r["n"], r["b"] + G["lb"], 0.0

    def g_eff(self):
        g = self.p.g.copy()
        for gk, st in self.gstate.items():
            G = self.p.groups[gk]
            if st["P"]:
                g += G["w"] * self.p.rows[st["pivot"]]["n"]
            if G["q"] > 0 and st["compl"]:
                # penalty w*xi + q/2*xi^2 with xi = n'a - b folded into the objective
                r = self.p.rows[G["rows"][0]]
                g += (G["w"] - G["q"] * r["b"]) * r["n"]
        return g

    def factor(self):
        Heff = self.p.H.copy()
        for gk, st in self.gstate.items():
            G = self.p.groups[gk]
            if G["q"] > 0 and st["compl"]:
                n = self.p.rows[G["rows"][0]]["n"]
                Heff += G["q"] * np.outer(n, n)
        self.Hinv = np.linalg.inv(Heff)
        m = len(self.W)
        if m == 0:
            self.C = np.zeros((0, self.N)); self.d = np.zeros(0); self.D = np.zeros(0)
            self.Pm = np.zeros((0, 0))
            return
        E = [self.eff(j) for j in self.W]
        self.C = np.array([e[0] for e in E]); self.d = np.array([e[1] for e in E])
        self.D = np.array([e[2] for e in E])
        S = self.C @ self.Hinv @ self.C.T + np.diag(self.D)
        self.Pm = np.linalg.inv(S)

    def refine(self, extra=None, rounds=2):
        """iterative refinement of the multipliers against the exactly evaluated residual of the
        working-set equations C a - D lam = d (the kernel does the same after every solve)"""
        if not len(self.W):
            return
        for _ in range(rounds):
            a = self.primal(extra)
            lam = np.array([self.lam[j] for j in self.W])
            res = self.C @ a - self.D * lam - self.d
            dl = self.Pm @ res
            for j, x in zip(self.W, dl):
                self.lam[j] += x

    def primal(self, extra=None):
        """a from the multipliers (extra = (normal, multiplier) of the incoming constraint)."""
        rhs = self.g_eff()
        if len(self.W):
            rhs = rhs + self.C.T @ np.array([self.lam[j] for j in self.W])
        if extra is not None:
            rhs = rhs + extra[1] * extra[0]
        return -self.Hinv @ rhs

    def solve_multipliers(self):
        """lambda for the current working set (used by warm start / refresh)."""
        self.factor()
        if len(self.W):
            lam = -self.Pm @ (self.d + self.C @ self.Hinv @ self.g_eff())
            for j, l in zip(self.W, lam):
                self.lam[j] = l

    # slack value of a group at point a
    def xi(self, g, a):
        G = self.p.groups[g]; st = self.gstate[g]
        if G["q"] > 0:
            if st["compl"]:
                r = self.p.rows[G["rows"][0]]
                return r["n"] @ a - r["b"]
            return G["lb"]
        if st["P"]:
            pr = self.p.rows[st["pivot"]]
            return pr["n"] @ a - pr["b"]
        return G["lb"]

    def group_margin(self, g, lam_q=None, q_in=None):
        """w - sum of explicit multipliers of the group (bound multiplier in Z, pivot
        multiplier in P); q_in: ('row', j) or ('bound', g) incoming with multiplier lam_q."""
        G = self.p.groups[g]
        s = sum(self.lam[j] for j in self.W if self.p.rows[j]["grp"] == g)
        if q_in is not None:
            if q_in[0] == "row" and self.p.rows[q_in[1]]["grp"] == g:
                s += lam_q
            if q_in[0] == "bound" and q_in[1] == g:
                s += lam_q
        return G["w"] - s

    # ------------------------------------------------------------------------------
    def most_violated(self, a):
        best, bq = self.tol, None
        inW = set(self.W)
        for j, r in enumerate(self.p.rows):
            if j in inW:
                continue
            g = r["grp"]
            if g is not None and self.gstate[g]["P"] and self.gstate[g]["pivot"] == j:
                continue
            if g is not None and self.p.groups[g]["q"] > 0 and self.gstate[g]["compl"]:
                continue
            val = r["n"] @ a - r["b"]
            if g is not None:
                val -= self.xi(g, a)
            sc = val / (1.0 + abs(r["b"]))
            if sc > best:
                best, bq = sc, ("row", j)
        for g, st in self.gstate.items():
            G = self.p.groups[g]
            if st["P"] or (G["q"] > 0 and st["compl"]):
                val = G["lb"] - self.xi(g, a)
                if val > best:
                    best, bq = val, ("bound", g)
        return bq

    def incoming_eff(self, q):
        """effective normal, rhs and compliance of the incoming constraint"""
        if q[0] == "row":
            j = q[1]
            r = self.p.rows[j]; g = r["grp"]
            if g is not None and self.p.groups[g]["q"] == 0 and self.gstate[g]["P"]:
                pr = self.p.rows[self.gstate[g]["pivot"]]
                return r["n"] - pr["n"], r["b"] - pr["b"], 0.0
            lb = self.p.groups[g]["lb"] if g is not None else 0.0
            return r["n"], r["b"] + lb, 0.0
        g = q[1]
        if self.p.groups[g]["q"] > 0:      # bound of a quadratic slack whose row is compliant
            pr = self.p.rows[self.p.groups[g]["rows"][0]]
            return -pr["n"], -(pr["b"] + self.p.groups[g]["lb"]), 0.0
        pr = self.p.rows[self.gstate[g]["pivot"]]
        return -pr["n"], -(pr["b"] + self.p.groups[g]["lb"]), 0.0

    # ------------------------------------------------------------------------------
    def solve(self, max_iter=2000, verbose=False):
        p = self.p
        if p.infeasible_const:
            self.status = 1
        self.status = 0
        self.factor()
        while True:
            if len(self.W):
                lam = -self.Pm @ (self.d + self.C @ (self.Hinv @ self.g_eff()))
                for j, l in zip(self.W, lam):
                    self.lam[j] = l
                if self.refine_rounds:
                    self.refine(None, self.refine_rounds)
            a = self.primal()
            q = self.most_violated(a)
            if q is None:
                break
            self.iters += 1
            if self.iters > max_iter:
                self.status = 2; break
            lam_q = 0.0
            done = False
            while not done:
                self.events += 1
                if self.events > 20 * max_iter:
                    self.status = 2; done = True; break
                c, dq, Dq = self.incoming_eff(q)
                if len(self.W):
                    lam = -self.Pm @ (self.d + self.C @ (self.Hinv @ (self.g_eff() + lam_q * c)))
                    for j, l in zip(self.W, lam):
                        self.lam[j] = l
                    if self.refine_rounds:
                        self.refine((c, lam_q), self.refine_rounds)
                a = self.primal((c, lam_q))
                viol = c @ a - Dq * lam_q - dq
                u = self.Hinv @ c
                m = len(self.W)
                s = self.C @ u if m else np.zeros(0)
                r = self.Pm @ s if m else np.zeros(0)
                zz = c @ u - (s @ r if m else 0.0) + Dq
                t2 = viol / zz if zz > 1e-8 * (c @ u + Dq) else math.inf
                if viol <= 0:
                    t2 = 0.0
                # blocking events
                t1, ev = math.inf, None
                for i, j in enumerate(self.W):
                    rj = p.rows[j]; g = rj["grp"]
                    lam = self.lam[j]
                    if g is not None and p.groups[g]["q"] > 0:
                        G = p.groups[g]
                        if r[i] > 0:
                            t = lam / r[i]
                            if t < t1: t1, ev = t, ("drop", j)
                        elif r[i] < 0:
                            t = (G["w"] - lam) / (-r[i])
                            if t < t1: t1, ev = t, ("compl", j)
                    else:
                        if r[i] > 0:
                            t = lam / r[i]
                            if t < t1: t1, ev = t, ("drop", j)
                if q[0] == "bound" and p.groups[q[1]]["q"] > 0:
                    # incoming slack bound of a penalised row: the row's own multiplier
                    # w + q*xi - mu must stay >= 0 (xi rises with the step as viol falls)
                    G = p.groups[q[1]]
                    xi_now = G["lb"] - viol
                    den = 1.0 - G["q"] * zz
                    if den > 0:
                        t = max(G["w"] + G["q"] * xi_now - lam_q, 0.0) / den
                        if t < t1: t1, ev = t, ("drop_h", q[1])
                for g, st in self.gstate.items():
                    G = p.groups[g]
                    if G["q"] > 0:
                        # Huber incoming row itself can hit its cap
                        if q[0] == "row" and p.rows[q[1]]["grp"] == g and not st["compl"]:
                            t = G["w"] - lam_q
                            if t < t1: t1, ev = t, ("cap_in", g)
                        continue
                    rate = -sum(r[i] for i, j in enumerate(self.W) if p.rows[j]["grp"] == g)
                    if (q[0] == "row" and p.rows[q[1]]["grp"] == g) or (q[0] == "bound" and q[1] == g):
                        rate += 1.0
                    if rate > 0:
                        t = max(self.group_margin(g, lam_q, q), 0.0) / rate
                        if t < t1: t1, ev = t, ("cap", g)
                t = min(t1, t2)
                if not math.isfinite(t):
                    self.status = 1; done = True; break
                for i, j in enumerate(self.W):
                    self.lam[j] -= t * r[i]
                lam_q += t
                if verbose:
                    print("  it", self.iters, "q", q, "viol", viol, "t", t, "ev", ev if t1 <= t2 else "full", "m", m)
                if t2 <= t1:
                    self.add_incoming(q, lam_q)
                    done = True
                else:
                    done = self.handle_event(ev, q, lam_q)
                self.factor()
        self.a = self.primal()
        return self.status

    # ------------------------------------------------------------------------------
    def add_incoming(self, q, lam_q):
        p = self.p
        if q[0] == "row":
            j = q[1]
            self.W.append(j); self.lam[j] = lam_q
            return
        g = q[1]
        if self.p.groups[g]["q"] > 0:
            # bound of a quadratic slack became active: the penalised row turns rigid
            # (its multiplier w + q*lb - mu is recomputed from the working set)
            self.gstate[g]["compl"] = False
            j = self.p.groups[g]["rows"][0]
            self.W.append(j); self.lam[j] = max(self.p.groups[g]["w"] - lam_q, 0.0)
            return
        # bound of a P group became active: group -> Z, pivot becomes an ordinary row
        st = self.gstate[g]
        piv = st["pivot"]
        lam_p = self.group_margin(g, lam_q, q)      # w - others - mu
        st["P"] = False; st["pivot"] = None
        self.W.append(piv); self.lam[piv] = lam_p

    def handle_event(self, ev, q, lam_q):
        """returns True if the incoming constraint is finished by this event."""
        p = self.p
        kind = ev[0]
        if kind == "drop":
            j = ev[1]
            self.W.remove(j); del self.lam[j]
            return False
        if kind == "compl":
            # multiplier reached the cap w: the slack leaves its bound, the row becomes a penalty
            j = ev[1]; g = p.rows[j]["grp"]
            self.W.remove(j); del self.lam[j]
            self.gstate[g]["compl"] = True
            return False
        if kind == "drop_h":
            # the incoming bound of a quadratic slack takes over: the row leaves, xi sits on its bound
            self.gstate[ev[1]]["compl"] = False
            return True
        if kind == "cap_in":
            # incoming row reached lambda = w before becoming tight: its slack leaves the bound and
            # the row turns into a penalty term of the objective; nothing is left to add
            self.gstate[ev[1]]["compl"] = True
            return True
        if kind == "cap":
            g = ev[1]; st = self.gstate[g]; G = p.groups[g]
            members = [j for j in self.W if p.rows[j]["grp"] == g]
            q_in_g = q[0] == "row" and p.rows[q[1]]["grp"] == g
            q_bound_g = q[0] == "bound" and q[1] == g
            if not st["P"]:
                # Z -> P
                if not members:
                    assert q_in_g
                    st["P"] = True; st["pivot"] = q[1]
                    return True
                piv = max(members, key=lambda j: self.lam[j])
                self.W.remove(piv); del self.lam[piv]
                st["P"] = True; st["pivot"] = piv
                return False
            # P: pivot multiplier reached zero
            if not members:
                if q_in_g:
                    st["pivot"] = q[1]
                    return True
                assert q_bound_g
                st["P"] = False; st["pivot"] = None
                return True
            piv = max(members, key=lambda j: self.lam[j])
            self.W.remove(piv); del self.lam[piv]
            st["pivot"] = piv
            return False
        raise RuntimeError(ev)

    # ------------------------------------------------------------------------------
    def slacks(self):
        a = self.a
        return {g: max(self.xi(g, a), self.p.groups[g]["lb"]) for g in self.p.groups}


def solve_ab_step(OPT, V, s0, v0, a_m1, t0, s_tv, v_tv, a_tv_prev, verbose=False):
    prob = ABProblem(OPT, V, s0, v0, a_m1, t0, s_tv, v_tv, a_tv_prev)
    qp = StructuredQP(prob)
    st = qp.solve(verbose=verbose)
    xi = qp.slacks()
    return prob, qp, st, xi


# ----------------------------------------------------------------------------------------
# warm start (receding-horizon shift of the previous step's working set)
def export_states(qp: StructuredQP):
    """row (name,k) -> code: 1 rigid in W, 2 pivot of a P group, 3 compliant"""
    st = {}
    p = qp.p
    for j in qp.W:
        r = p.rows[j]; g = r["grp"]
        code = 1
        if g is not None and p.groups[g]["q"] > 0 and qp.gstate[g]["compl"]:
            code = 3
        st[(r["name"], r["k"])] = code
    for g, s in qp.gstate.items():
        if s["P"]:
            r = p.rows[s["pivot"]]
            st[(r["name"], r["k"])] = 2
        if p.groups[g]["q"] > 0 and s["compl"]:
            r = p.rows[p.groups[g]["rows"][0]]
            st[(r["name"], r["k"])] = 3
    return st


def shift_states(st, N):
    out = {}
    for (name, k), code in st.items():
        if name.startswith("term"):
            out[(name, k)] = code
            continue
        if k >= 1:
            out[(name, k - 1)] = code
        if k == N - 1:
            out[(name, k)] = code
    return out


def warm_start(qp: StructuredQP, st, max_pass=16):
    p = qp.p
    index = {(r["name"], r["k"]): j for j, r in enumerate(p.rows)}
    for key, code in st.items():
        if key not in index:
            continue                      # constant rows (stage 0) are not in the a-space list
        j = index[key]; g = p.rows[j]["grp"]
        if code == 2:
            if g is None or p.groups[g]["q"] > 0 or qp.gstate[g]["P"]:
                continue
            qp.gstate[g]["P"] = True; qp.gstate[g]["pivot"] = j
    for key, code in st.items():
        if key not in index or code == 2:
            continue
        j = index[key]; g = p.rows[j]["grp"]
        if g is not None and qp.gstate[g]["P"] and qp.gstate[g]["pivot"] == j:
            continue
        if code == 3 and g is not None and p.groups[g]["q"] > 0:
            qp.gstate[g]["compl"] = True          # penalty row: not part of the working set
            continue
        qp.W.append(j)
    # drop linearly dependent rows (e.g. amax & jmax patterns) by rank check
    for it in range(max_pass):
        qp.lam = {}
        try:
            qp.solve_multipliers()
        except np.linalg.LinAlgError:
            qp.W = []; qp.lam = {}
            for g in qp.gstate.values():
                g.update(P=False, pivot=None, compl=False)
            qp.factor(); return False
        bad = False
        lam_max = max([abs(v) for v in qp.lam.values()] + [0.0])
        tol = qp.warm_tol * (1.0 + lam_max)
        # collect violations of dual feasibility, repair only the worst one per pass
        worst, fix = tol, None
        for j in list(qp.W):
            g = p.rows[j]["grp"]; lam = qp.lam[j]
            if g is not None and p.groups[g]["q"] > 0:
                G = p.groups[g]
                if -lam > worst: worst, fix = -lam, ("drop", j)
                if lam - G["w"] > worst: worst, fix = lam - G["w"], ("compl", j)
            elif -lam > worst:
                worst, fix = -lam, ("drop", j)
        for g, s_ in qp.gstate.items():
            G = p.groups[g]
            if G["q"] > 0:
                continue
            mg = qp.group_margin(g)
            if -mg > worst:
                worst, fix = -mg, ("cap", g)
        if fix is not None and it >= qp.single_passes:
            # mass repair: every violation at once (keeps the pivots, unlike a cold start)
            bad = True
            for j in list(qp.W):
                g = p.rows[j]["grp"]; lam = qp.lam[j]
                if g is not None and p.groups[g]["q"] > 0:
                    G = p.groups[g]
                    if lam < -tol: qp.W.remove(j)
                    elif lam > G["w"] + tol:
                        qp.W.remove(j); qp.gstate[g]["compl"] = True
                elif lam < -tol:
                    qp.W.remove(j)
            for g, s_ in qp.gstate.items():
                G = p.groups[g]
                if G["q"] > 0:
                    continue
                members = [j for j in qp.W if p.rows[j]["grp"] == g]
                mg = G["w"] - sum(max(qp.lam[j], 0.0) for j in members)
                if mg < -tol * (1 + G["w"]):
                    if members:
                        piv = max(members, key=lambda j: qp.lam[j])
                        qp.W.remove(piv); s_["P"] = True; s_["pivot"] = piv
                    else:
                        s_["P"] = False; s_["pivot"] = None
        elif fix is not None:
            bad = True
            if fix[0] == "drop":
                qp.W.remove(fix[1])
            elif fix[0] == "compl":
                qp.W.remove(fix[1])
                qp.gstate[p.rows[fix[1]]["grp"]]["compl"] = True
            else:
                g = fix[1]; s_ = qp.gstate[g]
                members = [j for j in qp.W if p.rows[j]["grp"] == g]
                if members:
                    piv = max(members, key=lambda j: qp.lam[j])
                    qp.W.remove(piv)
                    s_["P"] = True; s_["pivot"] = piv
                else:
                    s_["P"] = False; s_["pivot"] = None
        if not bad:
            qp.factor()
            return True
    qp.W = []; qp.lam = {}
    for g in qp.gstate.values():
        g.update(P=False, pivot=None, compl=False)
    qp.factor()
    return False
