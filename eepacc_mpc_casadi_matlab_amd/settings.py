"""Host-side mirror of the reference's configuration layer (the inputs of the hot path).

Same names and meaning as the reference so that a `Main.m` user finds the same knobs:

* ``SetVehicleParameters(tree)``  -- ABO/Functions/Settings/SetVehicleParameters.m:12-133
  (``tree='ABO'``: light commercial van; ``tree='ORIG'``: BMW i3 of
  ORIG/Functions/Settings/SetVehicleParameters.m).
* ``Settings(OPTsettings, tree)`` -- ABO/Settings.m:1-256 (custom use case 0 only; the canned
  use cases of GetUseCase.m are out of scope for now).
* ``GenerateUseCase`` / ``SimplifyPWA`` -- ABO/Functions/Settings/GenerateUseCase.m:50-116,
  ABO/Functions/PWA_function_manipulation/SimplifyPWA.m:14-49 (including its `doms(j)` quirk).
* ``Run_DrivingCycle`` -- ABO/Run_DrivingCycle.m:13-47 (lead-vehicle trace from a 10 Hz cycle).

Everything here is plain numpy on the host; it only prepares the plain-old-data config that
``eepacc_create`` receives.
"""
from __future__ import annotations

import math
from typing import Dict, Any

import numpy as np

__all__ = ["SetVehicleParameters", "Settings", "GenerateUseCase", "SimplifyPWA",
           "Run_DrivingCycle", "default_opt"]


def SetVehicleParameters(tree: str = "ABO") -> Dict[str, float]:
    V: Dict[str, float] = {}
    if tree == "ABO":      # ABO/.../SetVehicleParameters.m:12-19 ("DAILY")
        V.update(m=2620.0, A_f=4.614, c_d=0.46, L=3.52, h_g=1.2, WD_s_F=0.45)
        V.update(F0=275.0, F1=0.0, F2=1.305072)                       # :32-34
        V.update(p00=-1.178, p10=0.1154, p01=0.001764)                # :80-82
        V.update(c_r=0.0107, R_w=0.361)                               # :91-92
        # ICE fuel map and stepped gearbox (:44-46, :96-101), read by the ICE-map fuel term (OPT["fuel_map"] = "ICE")
        V.update(k00=-2.134, k10=0.01164, k01=0.01041, tau_fd=3.615, eta_drive=0.94 * 0.94)
        V["upSpd"] = np.array([15, 25, 30, 40, 55, 70, 85], dtype=np.float64) / (3.6 * 1.05)
        V["tau_gb"] = np.array([4.714, 3.314, 2.106, 1.667, 1.285, 1, 0.839, 0.667])
    elif tree == "ORIG":   # ORIG/.../SetVehicleParameters.m (BMW i3)
        V.update(m=1443.0, A_f=2.38, c_d=0.29, L=2.57, h_g=0.47, WD_s_F=0.53)
        V.update(F0=0.0, F1=0.0, F2=0.0, p00=0.0, p10=0.0, p01=0.0)
        V.update(c_r=0.0064, R_w=0.3498)
    else:
        raise ValueError("tree must be 'ABO' or 'ORIG'")
    V["L_f"] = V["WD_s_F"] * V["L"]
    V["L_r"] = V["L"] - V["L_f"]
    V["P_m_max"] = 125 * 1e3
    V["T_m_max"] = 250.0
    V["omega_m_r"] = 4800 / 60 * 2 * math.pi
    V["omega_m_max"] = 11400 / 60 * 2 * math.pi
    V["beta_gb"] = 9.665
    V["beta_fd"] = 1.0
    V["phi"] = V["beta_gb"] * V["beta_fd"] / V["R_w"]
    V["v_max"] = 150 / 3.6
    V["eta_gb"] = 0.985
    V["eta_fd"] = 0.93
    V["eta_TF"] = V["eta_gb"] * V["eta_fd"]
    V["lambda"] = 1.05
    V["mu"] = 0.8
    V["rho_a"] = 1.225
    V["g"] = 9.81
    V["zeta_a"] = .5 * V["c_d"] * V["rho_a"] * V["A_f"]
    return V


def SimplifyPWA(doms, vals):
    """ABO/Functions/PWA_function_manipulation/SimplifyPWA.m:14-49 (1-based loops restated)."""
    doms = [float(x) for x in doms]
    vals = [float(x) for x in vals]
    n = len(doms)
    doms_, vals_ = [doms[0]], [vals[0]]
    for i in range(1, n - 1):                                   # MATLAB i = 2:length-1
        with np.errstate(divide="ignore", invalid="ignore"):
            prevSlope = np.float64(vals[i] - vals[i - 1]) / np.float64(doms[i] - doms[i - 1])
            currSlope = np.float64(vals[i + 1] - vals[i]) / np.float64(doms[i + 1] - doms[i])
        if prevSlope != currSlope:
            doms_.append(doms[i])
            vals_.append(vals[i])
    doms2 = doms_ + [doms[-1]]
    vals2 = vals_ + [vals[-1]]
    domsNew, valsNew = [], []
    j = 0                                                        # MATLAB j = 1
    for i in range(len(doms2) - 1):
        if doms2[i] == doms2[i + 1]:
            if vals2[i] == vals2[i + 1]:
                pass
            else:
                domsNew.append(doms2[i] - .1)
                valsNew.append(vals2[i])
                j += 1
        else:
            domsNew.append(doms2[j])                            # sic: doms(j), :43-44
            valsNew.append(vals2[j])
            j += 1
    domsNew.append(doms2[-1])
    valsNew.append(vals2[-1])
    return np.array(domsNew), np.array(valsNew)


def GenerateUseCase(OPT: Dict[str, Any]) -> Dict[str, Any]:
    """ABO/Functions/Settings/GenerateUseCase.m:50-116."""
    def _empty(key):
        return key not in OPT or OPT[key] is None or np.size(OPT[key]) == 0
    if _empty("speedLimZones"):
        OPT["speedLimZones"] = np.array([[1e5, 0.0], [1e5, 1.0]])
    if _empty("curves"):
        OPT["curves"] = np.array([[1 / 1e5, 1.0, 2.0]])
    if _empty("slopes"):
        OPT["slopes"] = np.array([[0.0, 1.0, 2.0]])
    sRes = OPT["sRes"]
    Z = np.atleast_2d(np.asarray(OPT["speedLimZones"], dtype=np.float64))
    curves = np.atleast_2d(np.asarray(OPT["curves"], dtype=np.float64))
    slopes = np.atleast_2d(np.asarray(OPT["slopes"], dtype=np.float64))
    # maximum speed :53-71
    n = Z.shape[0]
    Z = Z[np.argsort(Z[:, 1], kind="stable")]
    Z = Z * np.array([1 / 3.6, 1.0])
    Z = np.vstack([Z, [Z[-1, 0], 1e5]])
    v_sl = np.zeros(2 * n + 1)
    s_sl = np.zeros(2 * n + 1)
    s_sl[0] = -1
    v_sl[0] = Z[0, 0]
    for i in range(1, 2 * n, 2):                                 # MATLAB i = 1:2:2n
        q = (i + 1) // 2                                         # 1-based zone index
        s_sl[i] = Z[q - 1, 1]
        v_sl[i] = Z[q - 1, 0]
        s_sl[i + 1] = Z[q, 1] - 1
        v_sl[i + 1] = Z[q - 1, 0]
    s_sl, v_sl = SimplifyPWA(s_sl, v_sl)
    # curvature :75-93
    curvature = np.zeros(int(np.max(curves[:, 2])))
    for i in range(curves.shape[0]):
        curvature[int(curves[i, 1]) - 1:int(curves[i, 2])] = abs(curves[i, 0])
    curvature[np.abs(curvature) <= 1e-6] = 1e-6
    s_curv = np.arange(1, len(curvature) + 1, sRes, dtype=np.float64)
    s_curv, curvature = SimplifyPWA(s_curv, curvature)
    s_curv = np.concatenate([s_curv, [s_curv[-1] + 1, s_curv[-1] + 2]])
    curvature = np.concatenate([curvature, [1e-6, 1e-6]])
    # slope :97-108
    slope = np.zeros(int(np.max(slopes[:, 2])))
    for i in range(slopes.shape[0]):
        slope[int(slopes[i, 1]) - 1:int(slopes[i, 2])] = slopes[i, 0]
    slope = np.arctan(slope / 100)
    s_slope = np.arange(1, len(slope) + 1, sRes, dtype=np.float64)
    s_slope, slope = SimplifyPWA(s_slope, slope)
    OPT.update(s_speedLim=s_sl, v_speedLim=v_sl, s_curv=s_curv, curvature=curvature,
               s_slope=s_slope, slope=slope)
    return OPT


def TimeVelToDist(t, v, s0: float, M: int) -> np.ndarray:
    """ABO/Functions/Other/TimeVelToDist.m:15-33: M-step forward Euler distance of a speed trace."""
    t = np.asarray(t, dtype=np.float64); v = np.asarray(v, dtype=np.float64)
    s = np.zeros(t.size)
    s[0] = s0
    for i in range(t.size - 1):
        DT = (t[i + 1] - t[i]) / M
        s_ = s[i]
        for _ in range(M):
            s_ = s_ + DT * v[i]
        s[i + 1] = s_
    return s


def GetUseCase(OPT: Dict[str, Any]) -> Dict[str, Any]:
    """ABO/Functions/Settings/GetUseCase.m:12-227: the predefined use cases (route tables, initial
    speed, simulated time, cut-off distance).  Cases 8 and 9 replay a recorded lead vehicle: the caller
    passes the two columns of the measurement file as OPT["argonne_lead"] = (t, v_mph)."""
    n = int(OPT["useCaseNum"])
    E3, E4 = np.zeros((0, 3)), np.zeros((0, 4))
    kmh = 1 / 3.6
    cases = {
        1: dict(speedLimZones=[[80, 0]], generateTVMPC=True, cutOffDist=300, t_sim=30),
        2: dict(v_init=80 * kmh, speedLimZones=[[80, 0]], stopLoc=[300], generateTVMPC=True, cutOffDist=300, t_sim=40),
        3: dict(v_init=80 * kmh, speedLimZones=[[80, 0], [120, 100], [80, 500], [30, 700], [80, 900]],
                generateTVMPC=True, cutOffDist=1200, t_sim=100),
        4: dict(v_init=0.0, speedLimZones=[[80, 0]], stopLoc=[500], generateTVMPC=True, cutOffDist=900, t_sim=75),
        5: dict(v_init=80 * kmh, speedLimZones=[[80, 0]],
                TLLoc=[[250, 9, 17, 12], [580, 18, 10, 15], [750, 1, 15, 10]], generateTVMPC=True, cutOffDist=850, t_sim=75),
        6: dict(speedLimZones=[[80, 0]], curves=[[-1 / 20, 100, 130], [1 / 40, 170, 230], [-1 / 80, 230, 250]],
                generateTVMPC=True, cutOffDist=400, t_sim=40),
        7: dict(v_init=58 * kmh, speedLimZones=[[60, 0]],
                slopes=[[0, 1, 300]] + [[k, 290 + 10 * k, 300 + 10 * k] for k in range(1, 9)],
                stopLoc=[600], generateTVMPC=True, cutOffDist=700, t_sim=70),
        10: dict(t_sim=60, speedLimZones=[[120, 0]], IncludeTV=False, generateTVMPC=False, TV_cuttingDist=30.0,
                 TV_cuttingVel=100 * kmh, v_init=120 * kmh, cutOffDist=1500),
        11: dict(v_init=0.0, speedLimZones=[[30, 0], [50, 600], [80, 1000], [120, 4450], [80, 10700]],
                 curves=[[-1 / 2, 200, 206], [-1 / 9, 2600, 2605], [1 / 17, 2605, 2650], [-1 / 12, 2650, 2655],
                         [-1 / 9, 3400, 3405], [1 / 17, 3405, 3450], [-1 / 12, 3450, 3455], [-1 / 30, 4300, 4310],
                         [-1 / 52, 4400, 4460], [-1 / 75, 7900, 8253]],
                 stopLoc=[80, 450, 600, 4000, 10700], cutOffDist=11.5e3, t_sim=700, generateTVMPC=False),
        12: dict(t_sim=100, v_init=0.0, speedLimZones=[[100, 0], [80, 300], [50, 510]], slopes=[[0, 200, 300]],
                 curves=[[-1 / 30, 220, 300], [1 / 20, 400, 430]], stopLoc=[350, 500],
                 TLLoc=[[450, 0, 15, 10], [250, 8, 15, 10]], cutOffDist=500, generateTVMPC=False),
    }
    cases[8] = dict(t_sim=110, speedLimZones=[[120, 0]], IncludeTV=True, generateTVMPC=False, cutOffDist=800)
    cases[9] = dict(t_sim=270, v_init=120 * kmh, speedLimZones=[[150, 0]], IncludeTV=True, generateTVMPC=False, cutOffDist=8e3)
    if n in (8, 9) and "argonne_lead" not in OPT:
        raise ValueError("use cases 8 and 9 replay the recorded lead vehicle of 'ArgonneData/61505019 Test Data.txt' "
                         "(GetUseCase.m:103-146): pass its time [s] / speed [mph] columns as OPT['argonne_lead'] = (t, v_mph)")
    if n not in cases:
        raise ValueError("unknown use case!")                               # GetUseCase.m:225
    uc = dict(slopes=E3, curves=E3, stopLoc=np.zeros(0), TLLoc=E4)
    uc.update(cases[n])
    for k in ("speedLimZones", "slopes", "curves", "TLLoc"):
        uc[k] = np.asarray(uc[k], dtype=np.float64).reshape(-1, 4 if k == "TLLoc" else (2 if k == "speedLimZones" else 3))
    uc["stopLoc"] = np.asarray(uc["stopLoc"], dtype=np.float64).ravel()
    OPT.update(uc)
    if n in (8, 9):                                                         # :103-146 recorded lead vehicle
        t_all, v_all = (np.asarray(x, dtype=np.float64) for x in OPT["argonne_lead"])
        t_start, gap = (4720.0, 20.0) if n == 8 else (4400.0, 50.0)
        use = (t_all >= t_start) & (t_all < t_start + OPT["t_sim"])
        t_tv, v_tv = t_all[use], v_all[use] * 0.44704
        s_tv = TimeVelToDist(t_tv, v_tv, OPT["s_init"] + gap, 5)
        f = int(round(OPT["Tvec"][0] / (t_tv[1] - t_tv[0])))
        OPT["s_tv"] = np.concatenate([s_tv[::f], s_tv[-1:]])
        OPT["v_tv"] = np.concatenate([v_tv[::f], v_tv[-1:]])
    if n == 10:                                                             # :148-163 lead vehicle cutting in
        Ts = OPT["Tvec"][0]
        k_tot = int(round(OPT["t_sim"] / Ts)) + 1
        OPT["s_tv"] = OPT["TV_cuttingDist"] + OPT["TV_cuttingVel"] * Ts * np.arange(k_tot)
        OPT["v_tv"] = OPT["TV_cuttingVel"] * np.ones(k_tot)
    return OPT


def default_opt() -> Dict[str, Any]:
    """The script-level settings of ABO/Main.m:22-53."""
    return dict(createGifs=False, IncludeTV=True, useCaseNum=0, Ts=0.5, t_sim=435.0)


def Settings(OPT: Dict[str, Any] | None = None, tree: str = "ABO", N_hor: int = 20) -> Dict[str, Any]:
    """ABO/Settings.m:1-256.  ``N_hor`` replaces the literal ``ones(1,20)`` of :101
    (BASELINE's N=30/60 configs change exactly that line)."""
    OPT = dict(default_opt() if OPT is None else OPT)
    OPT["tree"] = tree
    # NLP / FB weights :12-46
    W = [220.0, 0.3 * 3e5, 0.3 * 1e7, 4e5, 8e3, 1e7 * 9e0, 1e7]
    OPT["W_NLP"] = np.array(W)
    W = [50.0, 0.3 * 3e5, 0.3 * 1e7, 4e5, 8e3, 1e7 * 9e0, 1e7]
    OPT["W_FB"] = np.array(W)
    if tree == "ABO":      # :48-64 (7 weights, first = w_FC)
        OPT["W_AB"] = 1 * np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0])
    else:                  # ORIG/Settings.m:48-62 (6 weights)
        w_c, w_v, w_h, w_f = 0.1, 8e5, 1e4, 1e10
        OPT["W_AB"] = 1e-3 * np.array([w_c * 3e5, w_c * 1e7, w_v, w_h, w_f * 9e0, w_f])
    OPT["W_BL"] = np.array([1e2, 0.0, 0.0, 1e7])                             # :66-71 [w_v, w_a, w_j, w_f]
    Ts = OPT["Ts"]
    OPT.update(s_init=0.0, v_init=0 / 3.6, a_minus1=0.0)                     # :85-87
    OPT["FBuseTaylor"] = True                                                # :98
    moveBlockingSettings = np.ones(N_hor, dtype=int)                         # :101
    OPT["N_hor"] = int(moveBlockingSettings.sum())
    OPT.update(paramEstSetting=1, TVestSetting=1, tConstACC_ego=3.0, tConstACC_tar=5.0)  # :105-108
    OPT["N_integratePlant"] = 10                                             # :111
    OPT["solverToUse"] = 1                                                   # :114
    OPT["NLPmaxIter"] = 5000                                                 # :93
    OPT["Tvec"] = Ts * np.ones(OPT["N_hor"])                                 # :122
    OPT["s_goal"] = math.inf                                                 # :143
    OPT.update(sRes=1, tRes=1)
    if OPT["useCaseNum"] == 0:                                               # :150-193
        OPT.setdefault("slopes", np.zeros((0, 3)))
        OPT.setdefault("speedLimZones", np.array([[60.0, 0.0], [50.0, 1000.0]]))
        OPT.setdefault("curves", np.zeros((0, 3)))
        OPT.setdefault("stopLoc", np.zeros(0))
        OPT.setdefault("TLLoc", np.zeros((0, 4)))
        OPT.setdefault("cutOffDist", 3.5e3)
    else:
        OPT = GetUseCase(OPT)                                                # :195-198
    OPT.update(BL_a_LimLowVel=3.0, BL_a_LimHighVel=2.0, BL_j_LimLowVel=3.0, BL_j_LimHighVel=1.5,
               BL_N_hor=OPT["N_hor"], BL_Ts=Ts, BL_trajEstSett=1)            # :131-139
    OPT.update(h_min=2.0, tau_min=0.5)                                       # :203-204
    OPT.update(TVlength=4.0, TVinitDist=10.0, TVinitVel=0.0, TV_N_hor=20, TV_Ts=0.5)  # :207-212
    OPT.update(stopVel=0.2, stopRefDist=100.0, stopRefVelSlope=1.0, TLStopRegionSize=2.0,
               TLstopVel=-1.0, alpha_TTL=3.34)                               # :221-226
    OPT["b_quadr"] = np.array([185, 1.296e-20, 2.301, 0, 0.003728, -0.000181])  # :229
    OPT["b_fifthOrder"] = np.array([
        185, 0.427461350854152, 1.33881428409239, 0,
        0.00357911242725773, -0.000183195720362408, 0,
        2.41017657319644e-07, 1.92524236475911e-07,
        -1.21266753753542e-08, 2.69420315493954e-12,
        -6.73422331977247e-11, -4.65716485427471e-11,
        -3.29148609115376e-11, 8.05602619603684e-12,
        4.07099563902699e-16, 5.70507905296593e-15,
        4.78644673413403e-15, 2.49448524129528e-15,
        1.49184094719691e-15, -5.76405750523528e-16])                        # :230-238
    Mb = np.zeros(OPT["N_hor"], dtype=np.int32)                              # :243-250
    j = 0
    for n in moveBlockingSettings:
        Mb[j:j + n] = [0] + [1] * (n - 1)
        j += n
    OPT["Mb"] = Mb
    OPT["stopRefvelIncr"] = OPT["stopRefDist"] * OPT["stopRefVelSlope"]
    return GenerateUseCase(OPT)


def Settings_BL(OPT: Dict[str, Any]) -> Dict[str, Any]:
    """The view of OPTsettings that RunOpt_BLMPC takes (ABO/RunOpt_BLMPC.m:17-21, CreateQP_BL.m:26-33,
    EstimateVehicleTrajectory.m:25-29): horizon BL_N_hor with the uniform step BL_Ts, ego estimator BL_trajEstSett,
    weights W_BL, baseline comfort limits.  bl_mode = 1 selects CreateQP_BL behind the ABMPC entry points."""
    B = dict(OPT)
    N = int(OPT["BL_N_hor"])
    B.update(bl_mode=1, N_hor=N, Tvec=float(OPT["BL_Ts"]) * np.ones(N), Mb=np.zeros(N, dtype=np.int32),
             paramEstSetting=int(OPT["BL_trajEstSett"]))
    return B


def Run_DrivingCycle(OPT: Dict[str, Any], V_TO_10Hz: np.ndarray | None = None,
                     V_TO_resampled: np.ndarray | None = None):
    """ABO/Run_DrivingCycle.m:13-47.  Pass either the raw 10 Hz speed trace (resampled here
    like `resample(V_TO,1,5)`) or an already resampled one."""
    if V_TO_resampled is None:
        from scipy.signal import resample_poly
        V_TO = resample_poly(np.asarray(V_TO_10Hz, dtype=np.float64), 1, 5)   # :16
    else:
        V_TO = np.array(V_TO_resampled, dtype=np.float64)
    V_TO = np.where(V_TO < 0.1, 0.0, V_TO)                                   # :17
    Ts = OPT["TV_Ts"]
    n_cycle = int(round(OPT["t_sim"] / Ts))
    v = np.zeros(n_cycle + 1)
    s = np.zeros(n_cycle + 1)
    s[0] = OPT["TVinitDist"]
    for i in range(1, n_cycle):                                              # MATLAB i = 2:n_cycle
        v[i] = V_TO[i]
        s[i] = s[i - 1] + Ts * V_TO[i]
    v[n_cycle] = v[n_cycle - 1]                                              # :45-46
    s[n_cycle] = s[n_cycle - 1]
    return s, v
