"""ctypes mirror of include/eepacc.h (struct layouts and enum values only)."""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)

EEPACC_MAX_HORIZON = 63

OUT_FIELDS = ["s", "v", "Fm", "Fb", "a", "xi_v", "xi_h", "xi_s", "xi_f", "cost", "DistHor", "a_qp"]
OUT_N = len(OUT_FIELDS)
OUT = {name: i for i, name in enumerate(OUT_FIELDS)}

_VEH_FIELDS = ["m", "A_f", "c_d", "L", "h_g", "WD_s_F", "L_f", "L_r", "F0", "F1", "F2",
               "p00", "p10", "p01", "P_m_max", "T_m_max", "omega_m_r", "omega_m_max",
               "c_r", "R_w", "beta_gb", "beta_fd", "phi", "v_max", "eta_TF",
               "lambda", "mu", "rho_a", "g", "zeta_a",
               "k00", "k10", "k01", "tau_fd", "eta_drive"]
_VEH_OPTIONAL = {"k00": 0.0, "k10": 0.0, "k01": 0.0, "tau_fd": 1.0, "eta_drive": 1.0}


class Vehicle(C.Structure):
    _fields_ = ([(("lambda_" if f == "lambda" else f), C.c_double) for f in _VEH_FIELDS]
                + [("upSpd", C.c_double * 7), ("tau_gb", C.c_double * 8)])


class SettingsPOD(C.Structure):
    _fields_ = [
        ("N_hor", C.c_int32),
        ("Tvec", c_double_p),
        ("Mb", c_int32_p),
        ("W_AB", C.c_double * 7),
        ("W_FB", C.c_double * 7),
        ("ab_fuel_term", C.c_int32),
        ("ab_route_rows", C.c_int32),
        ("tau_min", C.c_double), ("h_min", C.c_double), ("s_goal", C.c_double),
        ("paramEstSetting", C.c_int32), ("TVestSetting", C.c_int32),
        ("tConstACC_ego", C.c_double), ("tConstACC_tar", C.c_double),
        ("N_integratePlant", C.c_int32),
        ("solverToUse", C.c_int32),
        ("FBuseTaylor", C.c_int32),
        ("b_quadr", C.c_double * 6),
        ("b_fifthOrder", C.c_double * 21),
        ("n_speedLim", C.c_int32), ("s_speedLim", c_double_p), ("v_speedLim", c_double_p),
        ("n_curv", C.c_int32), ("s_curv", c_double_p), ("curvature", c_double_p),
        ("n_slope", C.c_int32), ("s_slope", c_double_p), ("slope", c_double_p),
        ("n_stop", C.c_int32), ("stopLoc", c_double_p),
        ("n_TL", C.c_int32), ("TLLoc", c_double_p),
        ("stopRefDist", C.c_double), ("stopRefVelSlope", C.c_double), ("stopVel", C.c_double),
        ("TLstopVel", C.c_double), ("TLStopRegionSize", C.c_double), ("alpha_TTL", C.c_double),
        ("bl_mode", C.c_int32), ("bl_prox_iter", C.c_int32),
        ("W_BL", C.c_double * 4),
        ("BL_a_LimLowVel", C.c_double), ("BL_a_LimHighVel", C.c_double),
        ("BL_j_LimLowVel", C.c_double), ("BL_j_LimHighVel", C.c_double),
        ("bl_lp_eps", C.c_double),
        ("state_bound_tol", C.c_double),
    ]


def make_vehicle(V: Dict[str, float]) -> Vehicle:
    v = Vehicle()
    for f in _VEH_FIELDS:
        setattr(v, "lambda_" if f == "lambda" else f, float(V[f] if f not in _VEH_OPTIONAL else V.get(f, _VEH_OPTIONAL[f])))
    up = np.asarray(V.get("upSpd", np.full(7, 1e9)), dtype=np.float64).ravel()
    gb = np.asarray(V.get("tau_gb", np.ones(8)), dtype=np.float64).ravel()
    if up.size != 7 or gb.size != 8:
        raise ValueError("upSpd needs 7 and tau_gb 8 entries (SetVehicleParameters.m:96-98)")
    for i in range(7):
        v.upSpd[i] = float(up[i])
    for i in range(8):
        v.tau_gb[i] = float(gb[i])
    return v


class SettingsHolder:
    """Owns the numpy buffers the POD points into."""

    def __init__(self, OPT: Dict[str, Any]):
        self._keep = []
        self.pod = SettingsPOD()
        p = self.pod
        N = int(OPT["N_hor"])
        if N > EEPACC_MAX_HORIZON:
            raise ValueError(f"N_hor {N} exceeds EEPACC_MAX_HORIZON {EEPACC_MAX_HORIZON}")
        p.N_hor = N
        p.Tvec = self._dptr(OPT["Tvec"], N)
        mb = np.ascontiguousarray(OPT.get("Mb", np.zeros(N)), dtype=np.int32)
        self._keep.append(mb)
        p.Mb = mb.ctypes.data_as(c_int32_p)
        W_AB = np.asarray(OPT["W_AB"], dtype=np.float64).ravel()
        if W_AB.size == 7:
            p.ab_fuel_term = 2 if OPT.get("fuel_map", "EFF") == "ICE" else 1     # CreateQP_AB.m:154-166
            wab = W_AB
        elif W_AB.size == 6:       # ORIG/Settings.m:48-62: no w_FC entry
            p.ab_fuel_term = 0
            wab = np.concatenate([[0.0], W_AB])
        else:
            raise ValueError("W_AB must have 6 (ORIG) or 7 (ABO) entries")
        for i in range(7):
            p.W_AB[i] = float(wab[i])
            p.W_FB[i] = float(np.asarray(OPT["W_FB"]).ravel()[i])
        p.ab_route_rows = int(OPT.get("ab_route_rows", 1 if OPT.get("tree", "ABO") == "ORIG" else 0))
        p.tau_min = float(OPT["tau_min"]); p.h_min = float(OPT["h_min"]); p.s_goal = float(OPT["s_goal"])
        p.paramEstSetting = int(OPT["paramEstSetting"]); p.TVestSetting = int(OPT["TVestSetting"])
        p.tConstACC_ego = float(OPT["tConstACC_ego"]); p.tConstACC_tar = float(OPT["tConstACC_tar"])
        p.N_integratePlant = int(OPT["N_integratePlant"])
        p.solverToUse = int(OPT["solverToUse"])
        p.FBuseTaylor = int(bool(OPT["FBuseTaylor"]))
        for i in range(6):
            p.b_quadr[i] = float(OPT["b_quadr"][i])
        for i in range(21):
            p.b_fifthOrder[i] = float(OPT["b_fifthOrder"][i])
        p.n_speedLim = len(OPT["s_speedLim"])
        p.s_speedLim = self._dptr(OPT["s_speedLim"]); p.v_speedLim = self._dptr(OPT["v_speedLim"])
        p.n_curv = len(OPT["s_curv"])
        p.s_curv = self._dptr(OPT["s_curv"]); p.curvature = self._dptr(OPT["curvature"])
        p.n_slope = len(OPT["s_slope"])
        p.s_slope = self._dptr(OPT["s_slope"]); p.slope = self._dptr(OPT["slope"])
        stop = np.asarray(OPT.get("stopLoc", []), dtype=np.float64).ravel()
        p.n_stop = stop.size
        p.stopLoc = self._dptr(stop) if stop.size else c_double_p()
        TL = np.asarray(OPT.get("TLLoc", np.zeros((0, 4))), dtype=np.float64).reshape(-1, 4)
        p.n_TL = TL.shape[0]
        p.TLLoc = self._dptr(TL) if TL.size else c_double_p()
        p.stopRefDist = float(OPT["stopRefDist"]); p.stopRefVelSlope = float(OPT["stopRefVelSlope"])
        p.stopVel = float(OPT["stopVel"]); p.TLstopVel = float(OPT["TLstopVel"])
        p.TLStopRegionSize = float(OPT["TLStopRegionSize"]); p.alpha_TTL = float(OPT["alpha_TTL"])
        # baseline controller (RunOpt_BLMPC): settings.Settings_BL() fills these
        p.state_bound_tol = float(OPT.get("state_bound_tol", 0.0))
        p.bl_mode = int(OPT.get("bl_mode", 0))
        if p.bl_mode:
            for i in range(4):
                p.W_BL[i] = float(np.asarray(OPT["W_BL"]).ravel()[i])
            p.BL_a_LimLowVel = float(OPT["BL_a_LimLowVel"]); p.BL_a_LimHighVel = float(OPT["BL_a_LimHighVel"])
            p.BL_j_LimLowVel = float(OPT["BL_j_LimLowVel"]); p.BL_j_LimHighVel = float(OPT["BL_j_LimHighVel"])
            p.bl_lp_eps = float(OPT.get("bl_lp_eps", 0.0))
            p.bl_prox_iter = int(OPT.get("bl_prox_iter", 0))

    def _dptr(self, arr, n=None):
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        if n is not None and a.size != n:
            raise ValueError(f"expected {n} entries, got {a.size}")
        self._keep.append(a)
        return a.ctypes.data_as(c_double_p)


def as_dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def as_iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int32_p)
