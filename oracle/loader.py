"""ctypes front-end of oracle/_build/liboracle.so (see oracle/oracle.h)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Any, Dict

import numpy as np

from eepacc_mpc_casadi_matlab_amd._abi import (SettingsHolder, SettingsPOD, Vehicle, make_vehicle,
                                               OUT_N, EEPACC_MAX_HORIZON, c_double_p, as_dptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("qp_dense.c", "eepacc_oracle.c", "oracle.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "eepacc.h"))
    stale = force or not os.path.exists(_LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB


class QPStats(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("prox_iterations", C.c_int),
                ("n_active", C.c_int), ("polished", C.c_int),
                ("kkt_stationarity", C.c_double), ("kkt_primal", C.c_double),
                ("kkt_dual", C.c_double), ("rho", C.c_double)]


class LoopState(C.Structure):
    _fields_ = [("k", C.c_int), ("v_tv_measured", C.c_double),
                ("s_prev_sol", C.c_double * (EEPACC_MAX_HORIZON + 1)),
                ("v_prev_sol", C.c_double * (EEPACC_MAX_HORIZON + 1)),
                ("fbA22", C.c_double * EEPACC_MAX_HORIZON), ("fbD2", C.c_double * EEPACC_MAX_HORIZON),
                ("xwarm", C.c_double * (8 * EEPACC_MAX_HORIZON))]


class StepIO(C.Structure):
    _fields_ = [("s", C.c_double), ("v", C.c_double), ("a_prev", C.c_double), ("t0", C.c_double),
                ("s_tv", C.c_double), ("v_tv", C.c_double), ("a_tv_prev", C.c_double),
                ("v_prev", C.c_double), ("Fm_prev", C.c_double), ("Fb_prev", C.c_double),
                ("out", C.c_double * OUT_N),
                ("s_pred", C.c_double * (EEPACC_MAX_HORIZON + 1)),
                ("v_pred", C.c_double * (EEPACC_MAX_HORIZON + 1)),
                ("qp", QPStats)]


class Oracle:
    def __init__(self, OPT: Dict[str, Any], V: Dict[str, float]):
        self.lib = C.CDLL(build_oracle())
        self.OPT = OPT
        self.holder = SettingsHolder(OPT)
        self.S = self.holder.pod
        self.V = make_vehicle(V)
        self.N = int(OPT["N_hor"])
        L = self.lib
        L.orc_ab_num_rows.argtypes = [C.POINTER(SettingsPOD)]
        L.orc_fb_num_rows.argtypes = [C.POINTER(SettingsPOD)]
        L.orc_bl_num_rows.argtypes = [C.POINTER(SettingsPOD)]
        L.orc_ab_step.argtypes = [C.POINTER(SettingsPOD), C.POINTER(Vehicle), C.POINTER(StepIO), c_double_p]
        L.orc_fb_step.argtypes = [C.POINTER(SettingsPOD), C.POINTER(Vehicle), C.POINTER(LoopState),
                                  C.POINTER(StepIO), c_double_p]
        for f in (L.orc_run_abmpc, L.orc_run_fbmpc):
            f.argtypes = [C.POINTER(SettingsPOD), C.POINTER(Vehicle), C.c_int, C.c_double, C.c_double,
                          C.c_double, c_double_p, c_double_p, c_double_p, C.POINTER(C.c_int),
                          C.POINTER(C.c_int)]
        L.orc_run_plant_model.argtypes = [C.POINTER(SettingsPOD), C.POINTER(Vehicle), C.c_double,
                                          C.c_double, C.c_double, C.c_double, c_double_p, c_double_p]
        L.orc_postprocess.argtypes = [C.POINTER(SettingsPOD), C.POINTER(Vehicle), C.c_int, c_double_p,
                                      c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
        L.orc_qp_solve_dense.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                         c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                         C.c_double, C.c_int, c_double_p, c_double_p, C.POINTER(QPStats)]
        L.orc_estimate_vehicle_trajectory.argtypes = [C.POINTER(SettingsPOD), C.c_int, C.c_double,
                                                      C.c_double, C.c_double, c_double_p, c_double_p,
                                                      c_double_p, c_double_p]

    def lib_lut(self, v):
        """ABO/Functions/MPCs/LUTgearshift.m:17-41"""
        self.lib.orc_lut_gearshift.restype = C.c_double
        self.lib.orc_lut_gearshift.argtypes = [C.POINTER(Vehicle), C.c_double]
        return float(self.lib.orc_lut_gearshift(C.byref(self.V), float(v)))

    # sizes --------------------------------------------------------------------------------
    def nC(self, kind="ab"):
        if kind == "ab" and self.S.bl_mode:
            return int(self.lib.orc_bl_num_rows(C.byref(self.S)))
        f = self.lib.orc_ab_num_rows if kind == "ab" else self.lib.orc_fb_num_rows
        return int(f(C.byref(self.S)))

    def nV(self, kind="ab"):
        if kind == "ab" and self.S.bl_mode:
            return self.N * 2                      # baseline controller: a, xi_f
        return self.N * (5 if kind == "ab" else 6)

    # one open-loop step ----------------------------------------------------------------------
    def ab_step(self, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, want_dense=False):
        io = StepIO()
        io.s, io.v, io.a_prev, io.t0 = s, v, a_prev, t0
        io.s_tv, io.v_tv, io.a_tv_prev = s_tv, v_tv, a_tv_prev
        dense = None
        dptr = c_double_p()
        if want_dense:
            nV, nC = self.nV("ab"), self.nC("ab")
            dense = np.zeros(nV * nV + nV + nC * nV + 2 * nC + nV)
            dptr = as_dptr(dense)
        rc = self.lib.orc_ab_step(C.byref(self.S), C.byref(self.V), C.byref(io), dptr)
        return self._unpack(io, rc, dense, "ab")

    def fb_step(self, state: LoopState, s, v, v_prev, a_prev, Fm_prev, Fb_prev, t0, s_tv, v_tv,
                a_tv_prev, want_dense=False):
        io = StepIO()
        io.s, io.v, io.a_prev, io.t0 = s, v, a_prev, t0
        io.v_prev, io.Fm_prev, io.Fb_prev = v_prev, Fm_prev, Fb_prev
        io.s_tv, io.v_tv, io.a_tv_prev = s_tv, v_tv, a_tv_prev
        dense = None
        dptr = c_double_p()
        if want_dense:
            nV, nC = self.nV("fb"), self.nC("fb")
            dense = np.zeros(nV * nV + nV + nC * nV + 2 * nC + nV)
            dptr = as_dptr(dense)
        rc = self.lib.orc_fb_step(C.byref(self.S), C.byref(self.V), C.byref(state), C.byref(io), dptr)
        return self._unpack(io, rc, dense, "fb")

    def _unpack(self, io, rc, dense, kind):
        N = self.N
        res = dict(status=rc, out=np.array(io.out[:]), s_pred=np.array(io.s_pred[:N + 1]),
                   v_pred=np.array(io.v_pred[:N + 1]),
                   qp=dict((f, getattr(io.qp, f)) for f, _ in QPStats._fields_))
        if dense is not None:
            nV, nC = self.nV(kind), self.nC(kind)
            o = 0
            res["H"] = dense[o:o + nV * nV].reshape(nV, nV); o += nV * nV
            res["c"] = dense[o:o + nV]; o += nV
            res["G"] = dense[o:o + nC * nV].reshape(nC, nV); o += nC * nV
            res["lb"] = dense[o:o + nC]; o += nC
            res["ub"] = dense[o:o + nC]; o += nC
            res["x"] = dense[o:o + nV]
        return res

    # closed loop --------------------------------------------------------------------------
    def run(self, kind, n_steps, s0, v0, a_minus1, s_tv, v_tv):
        s_tv = np.ascontiguousarray(s_tv, dtype=np.float64)
        v_tv = np.ascontiguousarray(v_tv, dtype=np.float64)
        traj = np.zeros((n_steps, OUT_N))
        status = np.zeros(n_steps, dtype=np.int32)
        iters = np.zeros(n_steps, dtype=np.int32)
        f = self.lib.orc_run_abmpc if kind == "ab" else self.lib.orc_run_fbmpc
        f(C.byref(self.S), C.byref(self.V), n_steps, s0, v0, a_minus1, as_dptr(s_tv), as_dptr(v_tv),
          as_dptr(traj), status.ctypes.data_as(C.POINTER(C.c_int)), iters.ctypes.data_as(C.POINTER(C.c_int)))
        return traj, status, iters

    def plant(self, s, v, Fm, Fb):
        a = C.c_double(); b = C.c_double()
        self.lib.orc_run_plant_model(C.byref(self.S), C.byref(self.V), s, v, Fm, Fb, C.byref(a), C.byref(b))
        return a.value, b.value

    def postprocess(self, v, Fm):
        v = np.ascontiguousarray(v, dtype=np.float64); Fm = np.ascontiguousarray(Fm, dtype=np.float64)
        n = v.size
        rpm, Tm, P, E = (np.zeros(n) for _ in range(4))
        self.lib.orc_postprocess(C.byref(self.S), C.byref(self.V), n, as_dptr(v), as_dptr(Fm),
                                 as_dptr(rpm), as_dptr(Tm), as_dptr(P), as_dptr(E))
        return rpm, Tm, P, E

    def qp_solve(self, H, g, A, lba, uba, lbx=None, ubx=None, x0=None, rho_rel=0.0, max_prox=0):
        H = np.ascontiguousarray(H, dtype=np.float64); A = np.ascontiguousarray(A, dtype=np.float64)
        nV = H.shape[0]; nC = A.shape[0]
        g = np.ascontiguousarray(g, dtype=np.float64)
        def p(a):
            if a is None:
                return c_double_p()
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return as_dptr(a)
        keep = []
        x = np.zeros(nV); cost = C.c_double(); st = QPStats()
        self.lib.orc_qp_solve_dense(nV, nC, as_dptr(H), as_dptr(g), as_dptr(A), p(lba), p(uba),
                                    p(lbx), p(ubx), p(x0), rho_rel, max_prox, as_dptr(x),
                                    C.byref(cost), C.byref(st))
        return x, cost.value, dict((f, getattr(st, f)) for f, _ in QPStats._fields_)
