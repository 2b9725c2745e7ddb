"""Host side of RunOpt_NLP (ABO/RunOpt_NLP.m).

* ``build_tables`` -- the lookup tables RunOpt_NLP.m:63-184 builds before it formulates the problem (stops,
  traffic lights, the velocity-incentive profile through minPWA / SaturateSlopePWA / SimplifyPWA), as plain arrays.
* ``NlpEvaluator`` -- front-end of include/eepacc_nlp.h: objective, constraint rows, objective gradient and the
  integrator's Jacobian blocks of the multiple-shooting NLP for a batch of routes on the GPU (what IPOPT calls back
  into every iteration, RunOpt_NLP.m:505-510).  There is no CPU path: without the HIP library or a GPU it raises.
* ``NlpSolver`` / ``solve_routes`` / ``RunOpt_NLP`` -- the batched structured interior-point solver over the operators of
  include/eepacc_nlp.h (assembly, Riccati sweep, rollout, rows) and the cold-start multi-start; host logic only
  (per-route scalars and accept / reject masks), DESIGN.md section 3.8.
* ``postprocess`` -- RunOpt_NLP.m:545-605 (derived quantities and the cost series of optSol).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict

import numpy as np

from ._abi import Vehicle, make_vehicle

__all__ = ["pwa", "build_tables", "pick_start", "NlpEvaluator", "NlpSolver", "RunOpt_NLP", "car_following_start", "nlp_rows", "postprocess", "riccati_batched"]


def pwa(x, xs, ys):
    """Linear interpolation with linear extrapolation; returns (value, slope of the segment used)."""
    xs = np.asarray(xs, dtype=np.float64)
    ys = np.asarray(ys, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    i = np.clip(np.searchsorted(xs, x, side="right") - 1, 0, len(xs) - 2)
    dx = xs[i + 1] - xs[i]
    sl = np.where(dx > 0, (ys[i + 1] - ys[i]) / np.where(dx > 0, dx, 1.0), 0.0)
    return ys[i] + sl * (x - xs[i]), sl


def _interp_pwa(d, doms, vals):
    return float(np.interp(d, doms, vals))


def min_pwa(Ad, Av, Bd, Bv, simplify):
    """Functions/PWA_function_manipulation/minPWA.m: pointwise minimum of two PWA functions."""
    Ad, Av, Bd, Bv = (list(map(float, a)) for a in (Ad, Av, Bd, Bv))
    if Ad[0] != Bd[0]:
        if Ad[0] > Bd[0]:
            Ad, Av = [Bd[0]] + Ad, [Av[0]] + Av
        else:
            Bd, Bv = [Ad[0]] + Bd, [Bv[0]] + Bv
    if Ad[-1] != Bd[-1]:
        if Ad[-1] > Bd[-1]:
            Bd, Bv = Bd + [Ad[-1]], Bv + [Bv[-1]]
        else:
            Ad, Av = Ad + [Bd[-1]], Av + [Av[-1]]
    Ad, Av = (list(a) for a in simplify(np.array(Ad), np.array(Av)))
    Bd, Bv = (list(a) for a in simplify(np.array(Bd), np.array(Bv)))
    Ad, Av = Ad + [Ad[-1] + 1, Ad[-1] + 2], Av + [Av[-1], Av[-1]]
    Bd, Bv = Bd + [Bd[-1] + 1, Bd[-1] + 2], Bv + [Bv[-1], Bv[-1]]
    Cd, Cv = [], []
    iA = iB = 0
    doneA = doneB = False
    while True:
        Ad1, Av1, Ad2, Av2 = Ad[iA], Av[iA], Ad[iA + 1], Av[iA + 1]
        Bd1, Bv1, Bd2, Bv2 = Bd[iB], Bv[iB], Bd[iB + 1], Bv[iB + 1]
        As = (Av2 - Av1) / (Ad2 - Ad1)
        Bs = (Bv2 - Bv1) / (Bd2 - Bd1)
        if (Av1 > Bv1 and Av2 < Bv2) or (Av1 < Bv1 and Av2 > Bv2):
            s1 = (Bv1 - Av1 + (Ad1 - Bd1) * Bs) / (As - Bs)
            Id = Ad1 + s1
            if Ad1 <= Id <= Ad2 and Bd1 <= Id <= Bd2:
                Cd.append(Id)
                Cv.append(Av1 + As * s1)
        if Ad2 < Bd2:
            if Av1 <= _interp_pwa(Ad1, Bd, Bv):
                Cd.append(Ad1)
                Cv.append(Av1)
            iA += 1
            if iA == len(Ad) - 2:
                doneA = True
        else:
            if Bv1 <= _interp_pwa(Bd1, Ad, Av):
                Cd.append(Bd1)
                Cv.append(Bv1)
            iB += 1
            if iB == len(Bd) - 2:
                doneB = True
        if doneA and doneB:
            break
    order = np.argsort(np.array(Cd), kind="stable")
    return np.array(Cd)[order], np.array(Cv)[order]


def saturate_slope_pwa(doms, vals, c_des):
    """SaturateSlopePWA.m (+ FixCrossingPWA.m between its two passes)."""
    doms = np.array(doms, dtype=np.float64)
    vals = np.array(vals, dtype=np.float64)

    def one_pass():
        for i in range(1, len(doms)):
            c = (vals[i] - vals[i - 1]) / (doms[i] - doms[i - 1])
            if c > 0 and c > c_des:
                doms[i] = doms[i - 1] + (vals[i] - vals[i - 1]) / c_des
            elif c < 0 and c < -c_des:
                doms[i - 1] = doms[i] + (vals[i] - vals[i - 1]) / c_des
    one_pass()
    d0 = doms.copy()
    for cc in np.nonzero(np.diff(d0) <= 0)[0]:               # FixCrossingPWA.m (0-based cc = curCross-1)
        Ad1, Av1, Ad2, Av2 = doms[cc - 1], vals[cc - 1], doms[cc], vals[cc]
        Bd1, Bv1, Bd2, Bv2 = doms[cc + 1], vals[cc + 1], doms[cc + 2], vals[cc + 2]
        As = (Av2 - Av1) / (Ad2 - Ad1)
        Bs = (Bv2 - Bv1) / (Bd2 - Bd1)
        s1 = (Bv1 - Av1 + (Ad1 - Bd1) * Bs) / (As - Bs)
        Iv = Av1 + As * s1
        doms[cc], vals[cc] = d0[cc + 1], Iv
        doms[cc + 1], vals[cc + 1] = d0[cc], Iv
    one_pass()
    return doms, vals


def build_tables(OPT: Dict[str, Any]) -> Dict[str, Any]:
    """Lookup tables of RunOpt_NLP.m:63-184 as plain arrays."""
    from .settings import SimplifyPWA
    T: Dict[str, Any] = {}
    Ts = float(OPT["Ts"])
    N = int(round(float(OPT["t_sim"]) / Ts))
    T["slope"] = (np.asarray(OPT["s_slope"], float), np.asarray(OPT["slope"], float))
    T["flat"] = bool(np.sum(OPT["slope"]) < 1e-1)                                          # :363
    T["vlim"] = (np.asarray(OPT["s_speedLim"], float), np.asarray(OPT["v_speedLim"], float))
    T["curv"] = (np.asarray(OPT["s_curv"], float), np.asarray(OPT["curvature"], float))
    incr = float(OPT["stopRefDist"]) * float(OPT["stopRefVelSlope"])
    sS, vS = [], []
    for loc in np.sort(np.asarray(OPT["stopLoc"], float).ravel()):                         # :95-98
        sS += [loc - OPT["stopRefDist"], loc, loc + OPT["stopRefDist"]]
        vS += [incr, float(OPT["stopVel"]), incr]
    for i in range(len(vS) - 1):                                                           # :101-109
        if sS[i + 1] <= sS[i]:
            corr = .5 * (sS[i] - sS[i + 1]) + sS[i + 1]
            val = incr / (1 + OPT["stopRefDist"] / (sS[i] - sS[i + 1]))
            vS[i] = vS[i + 1] = val
            sS[i], sS[i + 1] = corr - 1, corr + 1
    if len(sS) < 1:
        sS, vS = [0.0, 1.0], [1e5, 1e5]                                                     # :112-115
    T["stop"] = (np.array(sS, float), np.array(vS, float))
    TL = np.asarray(OPT["TLLoc"], float).reshape(-1, 4) if np.size(OPT["TLLoc"]) else np.zeros((0, 4))
    T["tl_s"] = np.zeros((len(TL), 3))
    T["tl_v"] = np.array([incr, float(OPT["TLstopVel"]), incr])
    T["tl_state"] = np.zeros((len(TL), N))
    for i, row in enumerate(TL):                                                           # :128-156
        T["tl_s"][i] = [row[0] - OPT["stopRefDist"], row[0], row[0] + OPT["stopRefDist"]]
        for k in range(N):
            red = math.fmod(k * Ts - row[1], row[2] + row[3])
            if red < 0:
                red += row[2] + row[3]                                                     # MATLAB mod
            T["tl_state"][i, k] = .2 if red < row[2] else 1e3
    with np.errstate(divide="ignore"):
        vc = float(OPT["alpha_TTL"]) * np.abs(T["curv"][1]) ** (-1.0 / 3.0)
    sI, vI = min_pwa(T["vlim"][0], T["vlim"][1], T["curv"][0], vc, SimplifyPWA)           # :162
    sI, vI = saturate_slope_pwa(sI, vI, 0.5)                                               # :165
    keep = np.diff(sI) != 0                            # :168-172 ("~diff(s)==0" parses as (~diff(s))==0; short mask)
    sI, vI = np.append(sI[:-1][keep], sI[-1]), np.append(vI[:-1][keep], vI[-1])
    sI, vI = SimplifyPWA(sI, vI)                                                           # :175
    T["vinc"] = (np.asarray(sI, float), np.asarray(vI, float))
    T["N"] = N
    return T




# ----------------------------------------------------------------------------------------------
# front-end of include/eepacc_nlp.h
# ----------------------------------------------------------------------------------------------
class NlpProblemPOD(C.Structure):
    _fields_ = [("N", C.c_int32), ("n_tl", C.c_int32), ("flat", C.c_int32), ("pad", C.c_int32),
                ("Ts", C.c_double), ("W", C.c_double * 7), ("b", C.c_double * 21),
                ("s_goal", C.c_double), ("h_min", C.c_double), ("tau_min", C.c_double), ("alpha_TTL", C.c_double),
                ("n_vlim", C.c_int32), ("n_curv", C.c_int32), ("n_slope", C.c_int32), ("n_stop", C.c_int32),
                ("n_vinc", C.c_int32), ("pad2", C.c_int32),
                ("s_vlim", C.POINTER(C.c_double)), ("v_vlim", C.POINTER(C.c_double)),
                ("s_curv", C.POINTER(C.c_double)), ("curvature", C.POINTER(C.c_double)),
                ("s_slope", C.POINTER(C.c_double)), ("slope", C.POINTER(C.c_double)),
                ("s_stop", C.POINTER(C.c_double)), ("v_stop", C.POINTER(C.c_double)),
                ("s_vinc", C.POINTER(C.c_double)), ("v_vinc", C.POINTER(C.c_double)),
                ("tl_s", C.POINTER(C.c_double)), ("tl_v", C.c_double * 3), ("tl_state", C.POINTER(C.c_double))]


def nlp_rows(n_tl: int, s_goal: float) -> int:
    """Rows per interval (include/eepacc_nlp.h: eepacc_nlp_rows)."""
    return 17 + 2 * n_tl + 10 + (1 if math.isfinite(s_goal) else 0)


def _bind(lib):
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    lib.eepacc_nlp_rows.argtypes = [C.POINTER(NlpProblemPOD)]
    lib.eepacc_nlp_create.argtypes = [C.POINTER(vp), C.POINTER(NlpProblemPOD), C.POINTER(Vehicle), C.c_int]
    lib.eepacc_nlp_destroy.argtypes = [vp]
    lib.eepacc_nlp_destroy.restype = None
    lib.eepacc_nlp_eval.argtypes = [vp, C.c_int] + [vp] * 9
    lib.eepacc_nlp_synchronize.argtypes = [vp, vp]
    lib.eepacc_nlp_newton.argtypes = [vp, C.c_int, vp, C.c_double] + [vp] * 13
    lib.eepacc_nlp_rowdir.argtypes = [vp, C.c_int] + [vp] * 8
    lib.eepacc_nlp_rollout.argtypes = [vp, C.c_int] + [vp] * 7
    lib.eepacc_nlp_steprule.argtypes = [C.c_int, C.c_int, C.c_int] + [vp] * 10
    lib.eepacc_nlp_trial.argtypes = [C.c_int, C.c_int, C.c_int] + [vp] * 9
    lib.eepacc_nlp_riccati.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_double), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.eepacc_nlp_solve.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, C.POINTER(NlpOptions), vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int32), vp]
    lib.eepacc_run_nlp_host.argtypes = [vp, C.c_int, dp, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int32), dp, dp, C.POINTER(NlpOptions),
                                        dp, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, C.POINTER(C.c_int32)]
    lib.eepacc_nlp_car_following_start_host.argtypes = [vp, dp, C.c_double, C.c_double, C.c_int, C.c_double, dp]
    lib.eepacc_nlp_problem_from_settings.argtypes = [C.POINTER(vp), C.POINTER(NlpProblemPOD), vp, dp, dp, C.c_double, C.c_double]
    lib.eepacc_nlp_tables_free.argtypes = [vp]
    lib.eepacc_nlp_tables_free.restype = None
    lib.eepacc_nlp_postprocess_host.argtypes = [C.POINTER(Vehicle), dp, dp, C.c_double, C.c_int] + [dp] * 10
    return lib


class NlpOptions(C.Structure):
    """include/eepacc_nlp.h: eepacc_nlp_options (zero / negative entries select the library's defaults)"""
    _fields_ = [("max_iter", C.c_int32), ("restarts", C.c_int32), ("max_ls", C.c_int32), ("phase1_iter", C.c_int32),
                ("tol", C.c_double), ("mu_init", C.c_double), ("mu_min", C.c_double), ("obj_scale", C.c_double), ("margin", C.c_double),
                ("kink_eps_s", C.c_double), ("kink_eps_v", C.c_double)]


DEFAULT_STARTS = ((90, 4.0), (120, 2.0), (200, 2.0), (120, 4.0), (300, 2.0), (160, 8.0), (120, 8.0), (60, 2.0))


def tables_from_settings(OPTsettings: Dict[str, Any]) -> Dict[str, Any]:
    """The library's own table preprocessing (eepacc_nlp_problem_from_settings: RunOpt_NLP.m:63-184 in C++, no GPU) as the
    dict build_tables returns -- what a MEX gateway gets; the CPU tests compare it with an independent restatement."""
    from .engine import load_library, EepaccError
    from ._abi import SettingsHolder
    lib = _bind(load_library())
    O = dict(OPTsettings)
    O.setdefault("N_hor", 2); O.setdefault("Tvec", np.full(int(O["N_hor"]), float(O["Ts"])))
    holder = SettingsHolder(O)
    p = NlpProblemPOD()
    owner = C.c_void_p()
    W = (C.c_double * 7)(*[float(x) for x in np.asarray(OPTsettings["W_NLP"], float)])
    b = (C.c_double * 21)(*[float(x) for x in np.asarray(OPTsettings["b_fifthOrder"], float)])
    rc = lib.eepacc_nlp_problem_from_settings(C.byref(owner), C.byref(p), C.byref(holder.pod), W, b, float(OPTsettings["Ts"]), float(OPTsettings["t_sim"]))
    if rc != 0:
        raise EepaccError("eepacc_nlp_problem_from_settings failed (%d): %s" % (rc, lib.eepacc_last_error().decode()))
    try:
        arr = lambda ptr, n: np.array([ptr[i] for i in range(n)], dtype=np.float64)
        T = dict(N=int(p.N), flat=bool(p.flat), vlim=(arr(p.s_vlim, p.n_vlim), arr(p.v_vlim, p.n_vlim)),
                 curv=(arr(p.s_curv, p.n_curv), arr(p.curvature, p.n_curv)), slope=(arr(p.s_slope, p.n_slope), arr(p.slope, p.n_slope)),
                 stop=(arr(p.s_stop, p.n_stop), arr(p.v_stop, p.n_stop)), vinc=(arr(p.s_vinc, p.n_vinc), arr(p.v_vinc, p.n_vinc)),
                 tl_v=np.array(list(p.tl_v)), tl_s=arr(p.tl_s, 3 * p.n_tl).reshape(p.n_tl, 3) if p.n_tl else np.zeros((0, 3)),
                 tl_state=arr(p.tl_state, p.n_tl * p.N).reshape(p.n_tl, p.N) if p.n_tl else np.zeros((0, int(p.N))))
    finally:
        lib.eepacc_nlp_tables_free(owner)
    return T


class NlpEvaluator:
    """Batched nlp_f / nlp_g / nlp_grad_f / integrator Jacobian of RunOpt_NLP's problem on one GPU.

    ``OPTsettings`` as for RunOpt_NLP (t_sim, Ts, W_NLP, b_fifthOrder / b_quadr, useFifthOrderFit_NLP, the route
    tables of GenerateUseCase, h_min, tau_min, alpha_TTL, s_goal); ``V`` from SetVehicleParameters."""

    def __init__(self, OPTsettings: Dict[str, Any], V: Dict[str, float], device: int = 0):
        from .engine import load_library, EepaccError
        import torch
        self._err = EepaccError
        self.lib = _bind(load_library())
        if not torch.cuda.is_available():
            # still go through the library so that the failure is the library's own (no CPU path behind this class)
            pass
        T = build_tables(OPTsettings)
        self.tables = T
        self.N = int(T["N"])
        self.n_tl = int(T["tl_s"].shape[0])
        p = NlpProblemPOD()
        p.N, p.n_tl, p.flat = self.N, self.n_tl, int(T["flat"])
        p.Ts = float(OPTsettings["Ts"])
        p.W[:] = [float(x) for x in np.asarray(OPTsettings["W_NLP"], float)]
        if OPTsettings.get("useFifthOrderFit_NLP", True):
            b = np.asarray(OPTsettings["b_fifthOrder"], float)
        else:
            b = np.concatenate([np.asarray(OPTsettings["b_quadr"], float), np.zeros(15)])
        p.b[:] = [float(x) for x in b]
        p.s_goal, p.h_min = float(OPTsettings["s_goal"]), float(OPTsettings["h_min"])
        p.tau_min, p.alpha_TTL = float(OPTsettings["tau_min"]), float(OPTsettings["alpha_TTL"])
        self._keep = []

        def tab(key):
            xs = np.ascontiguousarray(T[key][0], dtype=np.float64)
            ys = np.ascontiguousarray(T[key][1], dtype=np.float64)
            self._keep += [xs, ys]
            return len(xs), xs.ctypes.data_as(C.POINTER(C.c_double)), ys.ctypes.data_as(C.POINTER(C.c_double))
        p.n_vlim, p.s_vlim, p.v_vlim = tab("vlim")
        p.n_curv, p.s_curv, p.curvature = tab("curv")
        p.n_slope, p.s_slope, p.slope = tab("slope")
        p.n_stop, p.s_stop, p.v_stop = tab("stop")
        p.n_vinc, p.s_vinc, p.v_vinc = tab("vinc")
        tls = np.ascontiguousarray(T["tl_s"], dtype=np.float64)
        tst = np.ascontiguousarray(T["tl_state"], dtype=np.float64)
        self._keep += [tls, tst]
        if self.n_tl:
            p.tl_s = tls.ctypes.data_as(C.POINTER(C.c_double))
            p.tl_state = tst.ctypes.data_as(C.POINTER(C.c_double))
        p.tl_v[:] = [float(x) for x in T["tl_v"]]
        self.pod = p
        self.R = int(self.lib.eepacc_nlp_rows(C.byref(p)))
        assert self.R == nlp_rows(self.n_tl, p.s_goal)
        self.V = make_vehicle(V)
        self.h = C.c_void_p()
        rc = self.lib.eepacc_nlp_create(C.byref(self.h), C.byref(p), C.byref(self.V), device)
        if rc != 0:
            raise EepaccError("eepacc_nlp_create failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        self.device = device

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            self.lib.eepacc_nlp_destroy(h)
            self.h = None

    def eval(self, s_tv, X, U, want_grad: bool = True, out=None):
        """s_tv [N][B], X [N+1][4][B], U [N][6][B] (torch CUDA tensors or numpy arrays, fp64).
        Returns dict(J [B], eq [N][4][B], ineq [N][R][B], gradJ [N][10][B], jacF [N][2][3][B]) of CUDA tensors."""
        import torch
        dev = torch.device("cuda", self.device)

        def d(x):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
            return t.to(device=dev, dtype=torch.float64).contiguous()
        s_tv, X, U = d(s_tv), d(X), d(U)
        N = self.N
        B = int(X.shape[-1])
        assert X.shape == (N + 1, 4, B) and U.shape == (N, 6, B) and s_tv.shape == (N, B)
        o = out or {}
        if "J" not in o:
            o["J"] = torch.empty(B, dtype=torch.float64, device=dev)
            o["eq"] = torch.empty((N, 4, B), dtype=torch.float64, device=dev)
            o["ineq"] = torch.empty((N, self.R, B), dtype=torch.float64, device=dev)
            if want_grad:
                o["gradJ"] = torch.empty((N, 10, B), dtype=torch.float64, device=dev)
                o["jacF"] = torch.empty((N, 2, 3, B), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        g = o["gradJ"].data_ptr() if want_grad else None
        jf = o["jacF"].data_ptr() if want_grad else None
        rc = self.lib.eepacc_nlp_eval(self.h, B, s_tv.data_ptr(), X.data_ptr(), U.data_ptr(), o["J"].data_ptr(),
                                      o["eq"].data_ptr(), o["ineq"].data_ptr(), g, jf, stream)
        if rc != 0:
            raise self._err("eepacc_nlp_eval failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return o

    def newton(self, s_tv, chi, u, lam, t, nu, mu: float, sigma: float):
        """eepacc_nlp_newton: route-major chi [B][N+1][4], u [B][N][6], lam / t [B][N][R], nu [B][N+1][4], s_tv [B][N].
        Returns (Q [B][N][10][10], q [B][N][10], AB [B][N][4][10], c [B][N][4], rows [B][N][R]) as CUDA tensors."""
        import torch
        dev = torch.device("cuda", self.device)

        def d(x):
            tt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
            return tt.to(device=dev, dtype=torch.float64).contiguous()
        s_tv, chi, u, lam, t, nu = (d(x) for x in (s_tv, chi, u, lam, t, nu))
        N, R = self.N, self.R
        B = int(chi.shape[0])
        assert chi.shape == (B, N + 1, 4) and u.shape == (B, N, 6) and lam.shape == (B, N, R) and t.shape == (B, N, R)
        assert nu.shape == (B, N + 1, 4) and s_tv.shape == (B, N)
        Q = torch.empty((B, N, 10, 10), dtype=torch.float64, device=dev)
        q = torch.empty((B, N, 10), dtype=torch.float64, device=dev)
        AB = torch.empty((B, N, 4, 10), dtype=torch.float64, device=dev)
        c = torch.empty((B, N, 4), dtype=torch.float64, device=dev)
        rows = torch.empty((B, N, R), dtype=torch.float64, device=dev)
        qlam = torch.empty((B, N, 10), dtype=torch.float64, device=dev)
        self.last_qlam = qlam
        stream = torch.cuda.current_stream(dev).cuda_stream
        mu_t = (mu if isinstance(mu, torch.Tensor) else torch.full((B,), float(mu), dtype=torch.float64)).to(device=dev, dtype=torch.float64).contiguous()
        rc = self.lib.eepacc_nlp_newton(self.h, B, mu_t.data_ptr(), float(sigma), s_tv.data_ptr(), chi.data_ptr(), u.data_ptr(),
                                        lam.data_ptr(), t.data_ptr(), nu.data_ptr(), Q.data_ptr(), q.data_ptr(), AB.data_ptr(),
                                        c.data_ptr(), rows.data_ptr(), qlam.data_ptr(), stream)
        if rc != 0:
            raise self._err("eepacc_nlp_newton failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return Q, q, AB, c, rows

    def synchronize(self):
        import torch
        stream = torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream
        rc = self.lib.eepacc_nlp_synchronize(self.h, stream)
        if rc != 0:
            raise self._err("eepacc_nlp_synchronize failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))


def riccati_batched(Q, q, AB, c, reg, reg_scale=(1e-6, 1e-6, 1e-10, 1e-10, 1e-10, 1e-10), device: int = 0, full: bool = False,
                    qlam=None):
    """include/eepacc_nlp.h: eepacc_nlp_riccati.  Q [B][N][10][10], q [B][N][10], AB [B][N][4][10], c [B][N][4], reg [B]
    (numpy or CUDA tensors).  Returns (dchi [B][N+1][4], du [B][N][6], nu [B][N+1][4], status [B]) as CUDA tensors."""
    import torch
    from .engine import load_library, EepaccError
    lib = _bind(load_library())
    dev = torch.device("cuda", device)

    def d(x):
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
        return t.to(device=dev, dtype=torch.float64).contiguous()
    Q, q, AB, c, reg = d(Q), d(q), d(AB), d(c), d(reg)
    B, N = int(Q.shape[0]), int(Q.shape[1])
    assert Q.shape == (B, N, 10, 10) and q.shape == (B, N, 10) and AB.shape == (B, N, 4, 10) and c.shape == (B, N, 4)
    assert reg.shape == (B,)
    dchi = torch.empty((B, N + 1, 4), dtype=torch.float64, device=dev)
    du = torch.empty((B, N, 6), dtype=torch.float64, device=dev)
    nu = torch.empty((B, N + 1, 4), dtype=torch.float64, device=dev)
    work = torch.empty((B, N, 50), dtype=torch.float64, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    gnorm = torch.empty(B, dtype=torch.float64, device=dev)
    sc = (C.c_double * 6)(*[float(x) for x in reg_scale])
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.eepacc_nlp_riccati(device, B, N, Q.data_ptr(), q.data_ptr(), AB.data_ptr(), c.data_ptr(), reg.data_ptr(), sc,
                                dchi.data_ptr(), du.data_ptr(), nu.data_ptr(), work.data_ptr(), status.data_ptr(), gnorm.data_ptr(),
                                qlam.data_ptr() if qlam is not None else None, stream)
    if rc != 0:
        raise EepaccError("eepacc_nlp_riccati failed (%d): %s" % (rc, lib.eepacc_last_error().decode()))
    if full:
        return dchi, du, nu, status, work, gnorm
    return dchi, du, nu, status


def postprocess(OPTsettings: Dict[str, Any], V: Dict[str, float], v_opt, Fm_opt, j_opt=None, slacks=None):
    """RunOpt_NLP.m:545-605: rpm, P (fifth-order surface), E, a, Tm and, given j_opt [N+1] and the slacks
    [N][4] = (xi_v, xi_h, xi_s, xi_f), the running cost series cost_P ... cost_xi_f of optSol."""
    Ts = float(OPTsettings["Ts"])
    v_opt, Fm_opt = np.asarray(v_opt, float), np.asarray(Fm_opt, float)
    rpm = (30 / math.pi) * v_opt[:-1] * V["phi"]
    b = np.asarray(OPTsettings["b_fifthOrder"], float)
    F, r = Fm_opt, rpm
    P = (b[0] + b[1] * F + b[2] * r + b[3] * F**2 + b[4] * F * r + b[5] * r**2 + b[6] * F**3 + b[7] * F**2 * r
         + b[8] * F * r**2 + b[9] * r**3 + b[10] * F**4 + b[11] * F**3 * r + b[12] * F**2 * r**2 + b[13] * F * r**3
         + b[14] * r**4 + b[15] * F**5 + b[16] * F**4 * r + b[17] * F**3 * r**2 + b[18] * F**2 * r**3
         + b[19] * F * r**4 + b[20] * r**5)                                     # GetMotorPower_FifthOrderSurface
    out = dict(rpm_opt=rpm, P_opt=P, E_opt=Ts * np.cumsum(P), a_opt=np.diff(v_opt) / Ts,
               Tm_opt=Fm_opt / V["phi"] / (V["eta_TF"] ** np.sign(Fm_opt)))
    if j_opt is not None and slacks is not None:
        W = np.asarray(OPTsettings["W_NLP"], float)
        sl = np.asarray(slacks, float)
        out.update(cost_P=W[0] * np.cumsum(P), cost_a=W[1] * np.cumsum(out["a_opt"] ** 2),
                   cost_j=W[2] * np.cumsum(np.asarray(j_opt, float)[:len(P)] ** 2),
                   cost_xi_v=W[3] * np.cumsum(sl[:, 0]), cost_xi_h=W[4] * np.cumsum(sl[:, 1]),
                   cost_xi_s=W[5] * np.cumsum(sl[:, 2]), cost_xi_f=W[6] * np.cumsum(sl[:, 3]))
    return out


# ----------------------------------------------------------------------------------------------
# interior-point iteration over the GPU operators (host logic only: step lengths, barrier and Levenberg updates)
# ----------------------------------------------------------------------------------------------
class NlpSolver(NlpEvaluator):
    """Batched structured interior-point solver of RunOpt_NLP's problem: every route of the batch runs the iteration of
    DESIGN.md section 3.8 (Newton system assembled by eepacc_nlp_newton, factorised by eepacc_nlp_riccati, closed-loop
    nonlinear forward pass eepacc_nlp_rollout, rows / objective by eepacc_nlp_rowdir / eepacc_nlp_eval); this class holds
    only the per-route scalars (barrier parameter, Levenberg term, step lengths) and the accept / reject logic, as
    elementwise tensor operations over the batch.  Convergence domain: see DESIGN.md section 7 (short routes from the
    car-following start; the full route from a start near the solution)."""

    REG_SCALE = (1e-6, 1e-6, 1e-10, 1e-10, 1e-10, 1e-10)

    def _theta(self, s):
        import torch
        if self.tables["flat"]:
            return torch.zeros_like(s)
        xs = torch.as_tensor(self.tables["slope"][0], dtype=torch.float64, device=s.device)
        ys = torch.as_tensor(self.tables["slope"][1], dtype=torch.float64, device=s.device)
        i = (torch.searchsorted(xs, s.contiguous(), right=True) - 1).clamp(0, len(xs) - 2)
        dx = xs[i + 1] - xs[i]
        sl = torch.where(dx > 0, (ys[i + 1] - ys[i]) / torch.where(dx > 0, dx, torch.ones_like(dx)), torch.zeros_like(dx))
        return ys[i] + sl * (s - xs[i])

    def _rowdir(self, s_tv, chi, u, dchi=None, du=None):
        import torch
        B = chi.shape[0]
        rows = torch.empty((B, self.N, self.R), dtype=torch.float64, device=chi.device)
        jdy = torch.empty_like(rows) if dchi is not None else None
        stream = torch.cuda.current_stream(chi.device).cuda_stream
        rc = self.lib.eepacc_nlp_rowdir(self.h, B, s_tv.data_ptr(), chi.data_ptr(), u.data_ptr(),
                                        dchi.data_ptr() if dchi is not None else None, du.data_ptr() if du is not None else None,
                                        rows.data_ptr(), jdy.data_ptr() if jdy is not None else None, stream)
        if rc != 0:
            raise self._err("eepacc_nlp_rowdir failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return rows, jdy

    def rollout(self, chi, u, work=None, alpha=None):
        import torch
        B = chi.shape[0]
        chi_n, u_n = torch.empty_like(chi), torch.empty_like(u)
        stream = torch.cuda.current_stream(chi.device).cuda_stream
        rc = self.lib.eepacc_nlp_rollout(self.h, B, alpha.data_ptr() if alpha is not None else None, chi.data_ptr(), u.data_ptr(),
                                         work.data_ptr() if work is not None else None, chi_n.data_ptr(), u_n.data_ptr(), stream)
        if rc != 0:
            raise self._err("eepacc_nlp_rollout failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return chi_n, u_n

    def _steprule(self, r, t, lam, jdy, mu, tau):
        """eepacc_nlp_steprule: (dt, dlam, out [B][8]); dt / dlam are None without jdy."""
        import torch
        B = r.shape[0]
        out = torch.empty((B, 8), dtype=torch.float64, device=r.device)
        dt = torch.empty_like(r) if jdy is not None else None
        dlam = torch.empty_like(r) if jdy is not None else None
        stream = torch.cuda.current_stream(r.device).cuda_stream
        rc = self.lib.eepacc_nlp_steprule(self.device, B, self.N * self.R, r.data_ptr(), t.data_ptr(), lam.data_ptr(),
                                          jdy.data_ptr() if jdy is not None else None, mu.data_ptr(), tau.data_ptr(),
                                          dt.data_ptr() if dt is not None else None, dlam.data_ptr() if dlam is not None else None,
                                          out.data_ptr(), stream)
        if rc != 0:
            raise self._err("eepacc_nlp_steprule failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return dt, dlam, out

    def _trial(self, r, t, dt, r_t, a, tau):
        """eepacc_nlp_trial: (t_trial, out [B][3] = feasible, residual sum, sum log t_trial)."""
        import torch
        B = r.shape[0]
        out = torch.empty((B, 3), dtype=torch.float64, device=r.device)
        t_t = torch.empty_like(r)
        stream = torch.cuda.current_stream(r.device).cuda_stream
        rc = self.lib.eepacc_nlp_trial(self.device, B, self.N * self.R, r.data_ptr(), t.data_ptr(), dt.data_ptr(), r_t.data_ptr(),
                                       a.data_ptr(), tau.data_ptr(), t_t.data_ptr(), out.data_ptr(), stream)
        if rc != 0:
            raise self._err("eepacc_nlp_trial failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return t_t, out

    def _values(self, s_tv, s_tv_bm, chi, u, sigma):
        """Scaled objective [B] and rows [B][N][R] at a point whose states are the rollout of its controls."""
        import torch
        X = torch.stack([chi[:, :, 0], chi[:, :, 1], self._theta(chi[:, :, 0]), chi[:, :, 3]], dim=2)      # [B][N+1][4]
        o = self.eval(s_tv_bm, X.permute(1, 2, 0).contiguous(), u.permute(1, 2, 0).contiguous(), want_grad=False)
        rows, _ = self._rowdir(s_tv, chi, u)
        return sigma * o["J"], rows

    def start_from_controls(self, s_tv, chi0, forces, margin=1.0):
        """Start point from force trajectories [B][N][2] (Fm, Fb <= 0): states by rollout, slacks `margin` above what the
        rows need.  chi0 [B][4] = (s_0, v_0, p_0, 0)."""
        import torch
        dev = torch.device("cuda", self.device)
        forces = torch.as_tensor(forces, dtype=torch.float64, device=dev)
        B, N = forces.shape[0], self.N
        u = torch.zeros((B, N, 6), dtype=torch.float64, device=dev)
        u[:, :, :2] = forces
        chi = torch.zeros((B, N + 1, 4), dtype=torch.float64, device=dev)
        chi[:, 0] = torch.as_tensor(chi0, dtype=torch.float64, device=dev)
        chi, u = self.rollout(chi, u)
        s_tv = torch.as_tensor(s_tv, dtype=torch.float64, device=dev).contiguous()
        r0, _ = self._rowdir(s_tv, chi, u)
        nt = 2 * self.n_tl
        zero = torch.zeros((), dtype=torch.float64, device=dev)
        u[:, :, 2] = torch.maximum(r0[:, :, 13 + nt], zero) + margin
        u[:, :, 3] = torch.maximum(r0[:, :, 16 + nt], zero) + margin
        u[:, :, 4] = torch.maximum(torch.maximum(r0[:, :, 12:13 + nt].amax(dim=2), r0[:, :, 15 + nt]), zero) + margin
        u[:, :, 5] = torch.maximum(r0[:, :, 0:12].amax(dim=2), zero) + margin
        return chi, u

    def solve_native(self, s_tv, chi0, forces, groups=None, max_iter=1500, mu_init=1.0, mu_min=1e-9, tol=1e-7, obj_scale=1e-5,
                     max_ls=4, restarts=3, margin=1.0, kink_eps_s=0.0, kink_eps_v=0.0):
        """eepacc_nlp_solve (include/eepacc_nlp.h): the whole interior-point iteration on the device -- no tensor operation
        and no host synchronisation inside an iteration.  s_tv [B][N], chi0 [B][4] = (s_0, v_0, p_0, 0), forces [B][N][2]
        (Fm, Fb <= 0: the start; states by rollout, slacks `margin` above what the rows need), groups [B] or None (starts of
        one problem share a group: the first KKT point ends it).  Returns dict(chi, u, J, status, iters, kkt, ticks)."""
        import torch
        dev = torch.device("cuda", self.device)
        f64 = torch.float64
        s_tv = torch.as_tensor(s_tv, dtype=f64, device=dev).contiguous()
        chi0 = torch.as_tensor(chi0, dtype=f64, device=dev).contiguous()
        forces = torch.as_tensor(forces, dtype=f64, device=dev).contiguous()
        B, N = int(forces.shape[0]), self.N
        assert s_tv.shape == (B, N) and chi0.shape == (B, 4) and forces.shape == (B, N, 2)
        n_groups = 0
        if groups is not None:
            groups = torch.as_tensor(groups, dtype=torch.int32, device=dev).contiguous()
            n_groups = int(groups.max().item()) + 1
        chi = torch.empty((B, N + 1, 4), dtype=f64, device=dev)
        u = torch.empty((B, N, 6), dtype=f64, device=dev)
        J = torch.empty(B, dtype=f64, device=dev)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        iters = torch.empty(B, dtype=torch.int32, device=dev)
        kkt = torch.empty((B, 6), dtype=f64, device=dev)
        opt = NlpOptions(max_iter=int(max_iter), restarts=int(restarts), max_ls=int(max_ls), tol=float(tol), mu_init=float(mu_init),
                         mu_min=float(mu_min), obj_scale=float(obj_scale), margin=float(margin), kink_eps_s=float(kink_eps_s),
                         kink_eps_v=float(kink_eps_v))
        ticks = C.c_int32(0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.eepacc_nlp_solve(self.h, B, s_tv.data_ptr(), groups.data_ptr() if groups is not None else None, n_groups,
                                       chi0.data_ptr(), forces.data_ptr(), C.byref(opt), chi.data_ptr(), u.data_ptr(), J.data_ptr(),
                                       status.data_ptr(), iters.data_ptr(), kkt.data_ptr(), C.byref(ticks), stream)
        if rc != 0:
            raise self._err("eepacc_nlp_solve failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return dict(chi=chi, u=u, J=J, status=status, iters=iters, kkt=kkt, ticks=int(ticks.value))

    def car_following_start_native(self, s_tv, s_init, v_init, lookahead, tau):
        """eepacc_nlp_car_following_start_host: the library's start generator (what eepacc_run_nlp_host uses); [N][2]."""
        s_tv = np.ascontiguousarray(s_tv, dtype=np.float64)
        out = np.zeros((self.N, 2))
        dp = C.POINTER(C.c_double)
        rc = self.lib.eepacc_nlp_car_following_start_host(self.h, s_tv.ctypes.data_as(dp), float(s_init), float(v_init), int(lookahead), float(tau),
                                                          out.ctypes.data_as(dp))
        if rc != 0:
            raise self._err("eepacc_nlp_car_following_start_host failed (%d)" % rc)
        return out

    def run_host(self, s_tv_routes, s_init, v_init, starts=DEFAULT_STARTS, start_forces=None, **opts):
        """eepacc_run_nlp_host: host arrays in, the optimum of every route out (the entry a MEX gateway calls)."""
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        stv = np.ascontiguousarray(np.asarray(s_tv_routes, dtype=np.float64)[:, :self.N])
        Rn, N = stv.shape[0], self.N
        S = 1 if start_forces is not None else len(starts)
        chi = np.zeros((Rn, N + 1, 4)); u = np.zeros((Rn, N, 6)); J = np.zeros(Rn)
        status = np.zeros(Rn, dtype=np.int32); iters = np.zeros(Rn, dtype=np.int32); start = np.zeros(Rn, dtype=np.int32)
        all_J = np.zeros((Rn, S)); all_st = np.zeros((Rn, S), dtype=np.int32)
        la = np.array([int(L) for (L, _) in starts], dtype=np.int32); tc = np.array([float(t) for (_, t) in starts], dtype=np.float64)
        sf = np.ascontiguousarray(start_forces, dtype=np.float64) if start_forces is not None else None
        opt = NlpOptions(**{k: v for k, v in opts.items() if v is not None})
        rc = self.lib.eepacc_run_nlp_host(self.h, Rn, stv.ctypes.data_as(dp), float(s_init), float(v_init), len(starts), la.ctypes.data_as(ip),
                                          tc.ctypes.data_as(dp), sf.ctypes.data_as(dp) if sf is not None else None, C.byref(opt),
                                          chi.ctypes.data_as(dp), u.ctypes.data_as(dp), J.ctypes.data_as(dp), status.ctypes.data_as(ip),
                                          iters.ctypes.data_as(ip), start.ctypes.data_as(ip), all_J.ctypes.data_as(dp), all_st.ctypes.data_as(ip))
        if rc != 0:
            raise self._err("eepacc_run_nlp_host failed (%d): %s" % (rc, self.lib.eepacc_last_error().decode()))
        return dict(chi=chi, u=u, J=J, status=status, iters=iters, start=start, all_J=all_J, all_status=all_st)

    def solve(self, s_tv, chi, u, max_iter=300, mu_init=1.0, mu_min=1e-9, tol=1e-7, obj_scale=1e-5, max_ls=4,
              reg_first=1e-4, reg_max=1e8, kappa_eps=10.0, kappa_mu=0.2, theta_mu=1.5, tau_min=0.99, verbose=False,
              until_first=False, groups=None, restarts=0, fused=True, accept_tol=1e-13):
        """s_tv [B][N], chi [B][N+1][4], u [B][N][6] (a start whose states are the rollout of its controls).
        Returns dict(chi, u, J [B], status [B] (0 KKT point to `tol`, 1 iteration limit, 2 Levenberg limit), iters [B],
        kkt [B][3], lam, t)."""
        import torch
        dev = torch.device("cuda", self.device)
        f64 = torch.float64
        s_tv = torch.as_tensor(s_tv, dtype=f64, device=dev).contiguous()
        s_tv_bm = s_tv.t().contiguous()
        chi = torch.as_tensor(chi, dtype=f64, device=dev).contiguous().clone()
        u = torch.as_tensor(u, dtype=f64, device=dev).contiguous().clone()
        B, N, R, sigma = chi.shape[0], self.N, self.R, float(obj_scale)
        cost, r = self._values(s_tv, s_tv_bm, chi, u, sigma)
        t = torch.clamp(-r, min=1e-2)
        mu = torch.full((B,), float(mu_init), dtype=f64, device=dev)
        lam = mu[:, None, None] / t
        nu = torch.zeros((B, N + 1, 4), dtype=f64, device=dev)
        rho = torch.ones(B, dtype=f64, device=dev)
        reg_last = torch.zeros(B, dtype=f64, device=dev)
        status = torch.ones(B, dtype=torch.int32, device=dev)
        iters = torch.zeros(B, dtype=torch.int32, device=dev)
        active = torch.ones(B, dtype=torch.bool, device=dev)
        kkt = torch.zeros((B, 3), dtype=f64, device=dev)
        inf = torch.full((B,), float("inf"), dtype=f64, device=dev)
        b3 = lambda x: x[:, None, None]
        n_restart = torch.zeros(B, dtype=torch.int32, device=dev)
        if until_first and groups is None:
            groups = torch.zeros(B, dtype=torch.int64, device=dev)
        if groups is not None:                              # starts of one problem share a group: the first KKT point ends it
            groups = torch.as_tensor(groups, dtype=torch.int64, device=dev)
            n_groups = int(groups.max().item()) + 1
        for it in range(max_iter):
            # Newton system at the current point; the barrier parameter falls while its subproblem is solved
            for _ in range(8):
                Q, q, AB, c, rows = self.newton(s_tv, chi, u, lam, t, nu, mu, sigma)
                rg = r + t
                dchi, du, nu_new, st, work, gnorm = riccati_batched(Q, q, AB, c, reg_last, self.REG_SCALE, self.device, full=True, qlam=self.last_qlam)
                e_dual = torch.where(st == 0, gnorm, inf)
                tau = torch.clamp(1.0 - mu, min=tau_min)
                if fused:
                    _, _, m8 = self._steprule(r, t, lam, None, mu, tau)
                    e_prim, e_comp0, e_compm = m8[:, 5], m8[:, 6], m8[:, 7]
                    rows_i = None
                else:
                    rows_i = rg > 1e-9 * (1.0 + t)
                    e_prim = (rg * rows_i).amax(dim=(1, 2))
                    lt = lam * t
                    e_comp0 = lt.amax(dim=(1, 2))
                    e_compm = (lt - b3(mu)).abs().amax(dim=(1, 2))
                err0 = torch.maximum(torch.maximum(e_dual, e_prim), e_comp0)
                done = active & (err0 <= tol)
                kkt = torch.where(active[:, None], torch.stack([e_dual, e_prim, e_comp0], dim=1), kkt)
                status = torch.where(done, torch.zeros_like(status), status)
                active = active & ~done
                dec = active & (mu > mu_min) & (torch.maximum(torch.maximum(e_dual, e_prim), e_compm) <= kappa_eps * mu)
                if not bool(dec.any()):
                    break
                mu = torch.where(dec, torch.clamp(torch.minimum(kappa_mu * mu, mu ** theta_mu), min=mu_min), mu)
            if verbose:
                print("it %3d active %d  J %s  mu %s  dual %s" % (it, int(active.sum()), (cost[:3] / sigma).tolist(), mu[:3].tolist(),
                                                                   e_dual[:3].tolist()), flush=True)
            if groups is not None:
                gdone = torch.zeros(n_groups, dtype=torch.int32, device=dev).scatter_reduce(0, groups, (status == 0).to(torch.int32), "amax")
                active = active & ~(gdone[groups] > 0)
            if not bool(active.any()):
                break
            iters += active.to(torch.int32)
            # Levenberg loop: the regularisation of a route grows until its factorisation has the right inertia and the
            # line search accepts a step
            reg = reg_last.clone()
            accepted = ~active
            skip = torch.zeros(B, dtype=torch.bool, device=dev)
            new_chi, new_u, new_t, new_cost, new_lam, new_nu = chi, u, t, cost, lam, nu
            ls_used = torch.zeros(B, dtype=torch.int32, device=dev)
            first = True
            for _attempt in range(40):
                need = active & ~accepted
                if not bool(need.any()):
                    break
                if not first:
                    dchi, du, nu_new, st, work, gnorm = riccati_batched(Q, q, AB, c, reg, self.REG_SCALE, self.device, full=True)
                first = False
                ok = st == 0
                _, jdy = self._rowdir(s_tv, chi, u, dchi, du)
                tau = torch.clamp(1.0 - mu, min=tau_min)
                if fused:
                    dt, dlam, m8 = self._steprule(r, t, lam, jdy, mu, tau)
                    a_p, a_d, infeas, lam_i, sumlog = m8[:, 0].contiguous(), m8[:, 1].contiguous(), m8[:, 2], m8[:, 3], m8[:, 4]
                else:
                    Dg = lam / t
                    dt = -rg - jdy
                    lam_new = b3(mu) / t + Dg * rg + Dg * jdy
                    dlam = lam_new - lam
                    big = torch.full_like(t, float("inf"))
                    a_p = torch.clamp(torch.where(dt < 0, -b3(tau) * t / torch.where(dt < 0, dt, -torch.ones_like(dt)), big).amin(dim=(1, 2)), max=1.0)
                    a_d = torch.clamp(torch.where(dlam < 0, -b3(tau) * lam / torch.where(dlam < 0, dlam, -torch.ones_like(dlam)), big).amin(dim=(1, 2)), max=1.0)
                    infeas = (rg * rows_i).sum(dim=(1, 2))
                    lam_i = (lam_new.abs() * rows_i).amax(dim=(1, 2))
                    sumlog = torch.log(t).sum(dim=(1, 2))
                lam_i = torch.nan_to_num(lam_i, nan=0.0, posinf=0.0)
                rho = torch.where((infeas > 0) & ok & need, torch.maximum(rho, 1.1 * lam_i), rho)   # only from a valid factorisation
                phi0 = cost - mu * sumlog + rho * infeas
                a = a_p.clone()
                acc_now = torch.zeros(B, dtype=torch.bool, device=dev)
                for ls in range(max_ls):
                    trial = need & ok & ~acc_now
                    if not bool(trial.any()):
                        break
                    chi_t, u_t = self.rollout(chi, u, work, a)
                    cost_t, r_t = self._values(s_tv, s_tv_bm, chi_t, u_t, sigma)
                    if fused:
                        t_t, m3 = self._trial(r, t, dt, r_t, a.contiguous(), tau)
                        feas, inf_t, sumlog_t = m3[:, 0] > 0.5, m3[:, 1], m3[:, 2]
                    else:
                        t_t = torch.where(rows_i, torch.maximum(-r_t, t + b3(a) * dt), -r_t)
                        feas = (t_t >= (1.0 - b3(tau)) * t).all(dim=2).all(dim=1)
                        inf_t = ((r_t + t_t) * rows_i).sum(dim=(1, 2))
                        sumlog_t = torch.log(torch.clamp(t_t, min=1e-300)).sum(dim=(1, 2))
                    phi_t = cost_t - mu * sumlog_t + rho * inf_t
                    good = trial & feas & (phi_t <= phi0 + accept_tol * phi0.abs())
                    g3 = b3(good)                                   # masked selects, no host round trip
                    new_chi, new_u, new_t = torch.where(g3, chi_t, new_chi), torch.where(g3, u_t, new_u), torch.where(g3, t_t, new_t)
                    new_cost = torch.where(good, cost_t, new_cost)
                    new_lam = torch.where(g3, lam + b3(a_d) * dlam, new_lam)
                    new_nu = torch.where(g3, nu_new, new_nu)
                    ls_used = torch.where(good, torch.full_like(ls_used, ls), ls_used)
                    acc_now = acc_now | good
                    a = torch.where(trial & ~good, 0.5 * a, a)
                accepted = accepted | acc_now
                grow = need & ~acc_now
                reg = torch.where(grow, torch.clamp(reg * 8.0, min=reg_first), reg)
                lost = grow & (reg > reg_max)
                if bool(lost.any()):
                    # no descent at any Levenberg term: the point is stationary for this barrier parameter up to a kink of
                    # the lookups.  While the barrier can still fall, lower it and go on; at its floor the route stops
                    cont = lost & (mu > mu_min)
                    mu = torch.where(cont, torch.clamp(kappa_mu * mu, min=mu_min), mu)
                    reg = torch.where(cont, torch.zeros_like(reg), reg)
                    skip = skip | cont
                    stop = lost & ~cont
                    if restarts > 0:
                        # crude restoration: re-centre the route at its current point (barrier and multipliers reset,
                        # slacks re-opened) a few times before giving it up
                        again = stop & (n_restart < restarts)
                        n_restart = n_restart + again.to(torch.int32)
                        mu = torch.where(again, torch.full_like(mu, float(mu_init)), mu)
                        t = torch.where(b3(again), torch.clamp(-r, min=1e-2), t)
                        lam = torch.where(b3(again), b3(mu) / t, lam)
                        new_t, new_lam = torch.where(b3(again), t, new_t), torch.where(b3(again), lam, new_lam)
                        rho = torch.where(again, torch.ones_like(rho), rho)
                        skip = skip | again
                        accepted = accepted | again
                        stop = stop & ~again
                    status = torch.where(stop, torch.full_like(status, 2), status)
                    active = active & ~stop
                    accepted = accepted | cont
            moved = accepted & active & ~skip
            if verbose:
                print("      reg %s a_p %s alpha %s a_d %s infeas %s rho %s ls %s moved %s" % (reg[:2].tolist(), a_p[:2].tolist(), a[:2].tolist(),
                      a_d[:2].tolist(), infeas[:2].tolist(), rho[:2].tolist(), ls_used[:2].tolist(), moved[:2].tolist()), flush=True)
            chi, u, t, cost, nu = new_chi, new_u, new_t, new_cost, new_nu
            lam = torch.where(b3(moved), torch.minimum(torch.maximum(new_lam, b3(mu) / (1e10 * t)), 1e10 * b3(mu) / t), lam)
            _, r = self._values(s_tv, s_tv_bm, chi, u, sigma)             # rows of the point every route now stands at
            reg_last = torch.where(moved, torch.where(ls_used <= 1, reg / 3.0, reg), reg_last)
            reg_last = torch.where((reg_last < reg_first) | skip, torch.zeros_like(reg_last), reg_last)
        return dict(chi=chi, u=u, J=cost / sigma, status=status, iters=iters, kkt=kkt, lam=lam, t=t)


def car_following_start(OPTsettings: Dict[str, Any], V: Dict[str, float], tables: Dict[str, Any], s_tv, lookahead=0,
                        tau=2.0) -> np.ndarray:
    """Force trajectory [N][2] (or [M][N][2] for M lead traces s_tv [M][N] with per-trace `lookahead` / `tau`) of a plain
    car-following rollout: speed target = min(speed limit - 1, stop profile, desired-headway speed behind the lead
    vehicle), acceleration (target - v)/tau clipped to [-2, 1.2] m/s^2 -- the start the solver is given where the reference
    starts IPOPT from z0 = 0 (RunOpt_NLP.m:348).  `lookahead` > 0 (samples) caps the target by the steady speed that
    reaches the lead vehicle's position that far ahead: a smooth cruise instead of stop and go."""
    s_tv = np.asarray(s_tv, dtype=np.float64)
    single = s_tv.ndim == 1
    stv = s_tv[None] if single else s_tv
    M_ = stv.shape[0]
    la = np.broadcast_to(np.asarray(lookahead, dtype=np.int64), (M_,))
    tc = np.broadcast_to(np.asarray(tau, dtype=np.float64), (M_,))
    N, Ts = int(tables["N"]), float(OPTsettings["Ts"])
    lm = V["lambda"] * V["m"]
    mg = V["m"] * V["g"]
    Fm_min = -V["phi"] * V["T_m_max"] / V["eta_TF"]
    s = np.full(M_, float(OPTsettings["s_init"]))
    v = np.full(M_, float(OPTsettings["v_init"]))
    out = np.zeros((M_, N, 2))
    flat = tables["flat"]
    rows = np.arange(M_)
    for k in range(N):
        th = np.zeros(M_) if flat else pwa(s, *tables["slope"])[0]
        grav = V["c_r"] * mg * np.cos(th) + mg * np.sin(th)
        vlim = pwa(s + 2.0 * v, *tables["vlim"])[0]
        stop = pwa(s + 2.0 * v, *tables["stop"])[0]
        gap = stv[:, min(k + 1, N - 1)] - 2.0 - 1.0 - s
        vt = np.maximum(0.0, np.minimum(np.minimum(vlim - 1.0, stop - 0.5), np.maximum(0.0, gap / 3.0)))
        kl = np.minimum(k + la, N - 1)
        ahead = np.maximum(0.0, (stv[rows, kl] - 3.0 - s) / ((kl - k) * Ts + 3.0))
        vt = np.where(la > 0, np.minimum(vt, ahead), vt)
        a = np.minimum(1.2, np.maximum(-2.0, (vt - v) / tc))
        a = np.where(v + a * Ts < 0.0, -v / Ts, a)
        F = lm * a + V["zeta_a"] * v * v + grav
        big = F > Fm_min * 0.5
        Fm = np.where(big, F, Fm_min * 0.5) + 1.0
        Fb = np.where(big, -1.0, F - Fm_min * 0.5)
        out[:, k, 0], out[:, k, 1] = Fm, Fb
        Ft = Fm + Fb
        DT = Ts / 4
        for _ in range(4):                                  # the interval's RK4 x 4 (RunOpt_NLP.m:262-278)
            a1 = (Ft - V["zeta_a"] * v * v - grav) / lm
            v2 = v + DT / 2 * a1
            a2 = (Ft - V["zeta_a"] * v2 * v2 - grav) / lm
            v3 = v + DT / 2 * a2
            a3 = (Ft - V["zeta_a"] * v3 * v3 - grav) / lm
            v4 = v + DT * a3
            a4 = (Ft - V["zeta_a"] * v4 * v4 - grav) / lm
            s = s + DT / 6 * (v + 2 * v2 + 2 * v3 + v4)
            v = v + DT / 6 * (a1 + 2 * a2 + 2 * a3 + a4)
    return out[0] if single else out


def pick_start(J, status, e_prim, feas_tol: float = 1e-6):
    """Winning start of every route, [R][S] tensors -> [R] indices, in tiers: the lowest objective among the starts at a
    KKT point (status 0); if a route has none, the lowest objective among its primal-feasible starts (constraint
    violation e_prim <= feas_tol); if none of those either, the smallest violation."""
    import torch
    inf = torch.full_like(J, float("inf"))
    Jn = torch.where(torch.isfinite(J), J, inf)
    tier1 = torch.where(status == 0, Jn, inf)
    tier2 = torch.where(e_prim <= feas_tol, Jn, inf)
    has1 = torch.isfinite(tier1).any(dim=1)
    has2 = torch.isfinite(tier2).any(dim=1)
    viol = torch.where(torch.isfinite(e_prim), e_prim, inf)
    return torch.where(has1, tier1.argmin(dim=1), torch.where(has2, tier2.argmin(dim=1), viol.argmin(dim=1)))


def solve_routes(sol: "NlpSolver", OPTsettings: Dict[str, Any], V: Dict[str, float], s_tv_routes, starts=DEFAULT_STARTS, max_iter: int | None = None,
                 native: bool = True, fused: bool = True, restarts: int = 3, kink_eps_s: float = 1e-2, phase1_iter: int = 1500):
    """Cold-start solve of R routes that share the route tables of `sol` and differ in their lead trace [R][N]: every route
    gets the multi-start of RunOpt_NLP (len(starts) instances, one group), all R * S instances run as one batch through
    eepacc_nlp_solve (native = False: the round-2 host loop over the single operators, kept as a cross-check).  max_iter:
    OPTsettings['NLPmaxIter'] (RunOpt_NLP.m:247) when not given.  Returns per route: J, status, iterations, index of the
    winning start, chi [R][N+1][4], u [R][N][6]."""
    import torch
    N = sol.N
    if max_iter is None:
        max_iter = int(OPTsettings.get("NLPmaxIter", 5000))
    s_tv_routes = np.asarray(s_tv_routes, dtype=np.float64)[:, :N]
    Rn, S = s_tv_routes.shape[0], len(starts)
    s0, v0 = float(OPTsettings["s_init"]), float(OPTsettings["v_init"])
    th0 = 0.0 if sol.tables["flat"] else float(pwa(s0, *sol.tables["slope"])[0])
    p0 = -(V["zeta_a"] * v0 * v0 + V["c_r"] * V["m"] * V["g"] * math.cos(th0) + V["m"] * V["g"] * math.sin(th0)) / (V["lambda"] * V["m"])
    stv = np.repeat(s_tv_routes, S, axis=0)
    forces = car_following_start(OPTsettings, V, sol.tables, stv, lookahead=np.tile([min(int(L), N - 1) for (L, _) in starts], Rn),
                                 tau=np.tile([float(tc) for (_, tc) in starts], Rn))
    groups = np.repeat(np.arange(Rn), S)
    chi0 = np.tile(np.array([[s0, v0, p0, 0.0]]), (Rn * S, 1))
    two_phase = native and kink_eps_s > 0
    it1 = min(max_iter, phase1_iter) if two_phase else max_iter          # as eepacc_run_nlp_host: first phase on the exact tables
    if native:
        R = sol.solve_native(stv, chi0, forces, groups=groups, max_iter=it1, mu_init=1.0, restarts=restarts, margin=1.0)
    else:
        chi, u = sol.start_from_controls(stv, chi0, forces, margin=1.0)
        R = sol.solve(stv, chi, u, max_iter=max_iter, mu_init=1.0, groups=groups, fused=fused, restarts=restarts)
    st = R["status"].view(Rn, S)
    win = pick_start(R["J"].view(Rn, S), st, R["kkt"][:, 1].view(Rn, S))
    idx = torch.arange(Rn, device=win.device) * S + win
    out = dict(J=R["J"][idx].clone(), status=R["status"][idx].clone(), iters=R["iters"][idx].clone(), start=win, chi=R["chi"][idx].clone(),
               u=R["u"][idx].clone(), all_J=R["J"].view(Rn, S), all_status=st, all_kkt=R["kkt"].view(Rn, S, -1), all_iters=R["iters"].view(Rn, S),
               ticks=R.get("ticks"), second_phase=torch.zeros(Rn, dtype=torch.bool, device=win.device))
    bad = torch.nonzero(out["status"] != 0).flatten()
    if native and kink_eps_s > 0 and bad.numel():
        # second phase (as in eepacc_run_nlp_host): routes none of whose starts reached a KKT point -- the minimiser pins a
        # node on a table knot -- again from the forces of their best start, with the kinks of the position tables rounded
        f2 = out["u"][bad][:, :, :2].clone()
        f2[:, :, 1] = torch.clamp(f2[:, :, 1], max=-1e-3)
        R2 = sol.solve_native(torch.as_tensor(s_tv_routes, device=f2.device)[bad], chi0[:bad.numel()], f2, max_iter=max(max_iter - it1, 1500), mu_init=1e-2,
                              restarts=restarts, margin=1e-1, kink_eps_s=kink_eps_s)
        ok = R2["status"] == 0
        tgt = bad[ok]
        out["J"][tgt] = R2["J"][ok]; out["status"][tgt] = 0; out["iters"][tgt] += R2["iters"][ok]
        out["chi"][tgt] = R2["chi"][ok]; out["u"][tgt] = R2["u"][ok]; out["second_phase"][tgt] = True
    return out


def RunOpt_NLP(OPTsettings: Dict[str, Any], V: Dict[str, float] | None = None, device: int = 0, start_forces=None,
               max_iter: int | None = None, mu_init: float | None = None, starts=DEFAULT_STARTS) -> Dict[str, Any]:
    """`optSol = RunOpt_NLP(OPTsettings)` (ABO/RunOpt_NLP.m, called from ABO/Main.m:97): same fields as the reference's
    struct, through eepacc_run_nlp_host -- the entry a MEX gateway calls (mex/RunOpt_NLP.c).  `OPTsettings["s_tv"]` is the
    lead trace of Main.m:88; `OPTsettings["NLPmaxIter"]` the iteration limit of RunOpt_NLP.m:247.

    The problem has many local solutions of nearly equal objective (a stop-and-go trajectory behind the lead vehicle is
    0.5 % above the smooth cruise IPOPT finds from z0 = 0), so the cold start is a **multi-start in one batch**:
    car-following rollouts with different look-ahead horizons / response times (`starts` = (look-ahead samples, time
    constant)) run side by side until the first reaches a KKT point; if none does, the best start by
    eepacc_run_nlp_host's tiers is returned with `exitMessage` 'Maximum_Iterations_Exceeded' / 'Restoration_Failed'
    (IPOPT's names for those outcomes).  `start_forces` [N][2] (Fm, Fb) replaces the multi-start by one warm start."""
    import time
    from .settings import SetVehicleParameters
    V = V or SetVehicleParameters(OPTsettings.get("tree", "ABO"))
    sol = NlpSolver(OPTsettings, V, device=device)
    N = sol.N
    s_tv = np.asarray(OPTsettings["s_tv"], dtype=np.float64)[:N]
    s0, v0 = float(OPTsettings["s_init"]), float(OPTsettings["v_init"])
    if max_iter is None:
        max_iter = int(OPTsettings.get("NLPmaxIter", 5000))
    t0 = time.perf_counter()
    sf = None if start_forces is None else np.asarray(start_forces, float)[None]
    R = sol.run_host(s_tv[None], s0, v0, starts=starts, start_forces=sf, max_iter=max_iter, mu_init=mu_init)
    tSolve = time.perf_counter() - t0
    chi, u, i = R["chi"][0], R["u"][0], int(R["start"][0])
    st, Jb = R["all_status"][0], R["all_J"][0]
    theta = np.zeros(N + 1) if sol.tables["flat"] else pwa(chi[:, 0], *sol.tables["slope"])[0]
    out = dict(s_velInc=sol.tables["vinc"][0], v_velInc=sol.tables["vinc"][1], tSolve=tSolve,
               exitMessage={0: "Solve_Succeeded", 1: "Maximum_Iterations_Exceeded", 2: "Restoration_Failed"}[int(R["status"][0])],
               s_opt=chi[:, 0], v_opt=chi[:, 1], theta_opt=theta, j_opt=chi[:, 3], Fm_opt=u[:, 0], Fb_opt=u[:, 1],
               xi_v_opt=u[:, 2], xi_h_opt=u[:, 3], xi_s_opt=u[:, 4], xi_f_opt=u[:, 5],
               J=float(R["J"][0]), iterations=int(R["iters"][0]), start_index=i, starts_J=Jb.tolist(), starts_status=st.tolist())
    out.update(postprocess(OPTsettings, V, chi[:, 1], u[:, 0], chi[:, 3], u[:, 2:]))
    return out
