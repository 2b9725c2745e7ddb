"""Host-side mirror of the reference's operator interface over the C-ABI of libeepacc.

``RunOpt_ABMPC(OPTsettings)`` (ABO/RunOpt_ABMPC.m:1) keeps its name and the fields of its
``optSol`` result; ``Engine`` is the batched form (B independent ego/scenario instances, one
wavefront each).  torch is used only for device memory and streams; every compute call goes
through ``libeepacc.so``.  There is no CPU fallback: a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, Optional

import numpy as np

from ._abi import SettingsHolder, SettingsPOD, Vehicle, make_vehicle, OUT, OUT_N, OUT_FIELDS, c_double_p, as_dptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libeepacc.so")
_lib = None


class EepaccError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load libeepacc.so (built in-tree by build.py).  Fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  -- before the CDLL: libeepacc must bind to the HIP runtime torch ships, not load a second one
    except ImportError:
        pass
    if not os.path.exists(_LIBPATH):
        raise EepaccError(f"{_LIBPATH} not found: build it with `python -m eepacc_mpc_casadi_matlab_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(_LIBPATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p
    lib.eepacc_last_error.restype = C.c_char_p
    lib.eepacc_version.restype = C.c_int
    lib.eepacc_sizeof_settings.restype = C.c_int
    lib.eepacc_sizeof_vehicle.restype = C.c_int
    if lib.eepacc_sizeof_settings() != C.sizeof(SettingsPOD) or lib.eepacc_sizeof_vehicle() != C.sizeof(Vehicle):
        raise EepaccError("ctypes mirror of include/eepacc.h is out of date (struct size mismatch)")
    lib.eepacc_create.argtypes = [C.POINTER(vp), C.POINTER(SettingsPOD), C.POINTER(Vehicle), C.c_int, C.c_int]
    lib.eepacc_destroy.argtypes = [vp]
    lib.eepacc_destroy.restype = None
    lib.eepacc_reset.argtypes = [vp]
    lib.eepacc_ab_step.argtypes = [vp, C.c_int] + [dp] * 7 + [dp, dp, dp, dp, vp]
    lib.eepacc_run_abmpc.argtypes = [vp, C.c_int, C.c_int] + [dp] * 5 + [dp, dp, vp]
    lib.eepacc_run_abmpc_host.argtypes = [vp, C.c_int, C.c_int] + [c_double_p] * 5 + [c_double_p, ip]
    lib.eepacc_bl_step.argtypes = lib.eepacc_ab_step.argtypes
    lib.eepacc_run_blmpc.argtypes = lib.eepacc_run_abmpc.argtypes
    lib.eepacc_run_blmpc_host.argtypes = lib.eepacc_run_abmpc_host.argtypes
    lib.eepacc_postprocess.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, vp]
    lib.eepacc_last_iterations.argtypes = [vp, C.c_int, ip]
    lib.eepacc_fb_step.argtypes = [vp, C.c_int] + [dp] * 10 + [dp, dp, dp, dp, vp]
    lib.eepacc_run_fbmpc.argtypes = [vp, C.c_int, C.c_int] + [dp] * 5 + [dp, dp, vp]
    lib.eepacc_run_fbmpc_host.argtypes = [vp, C.c_int, C.c_int] + [c_double_p] * 5 + [c_double_p, ip]
    lib.eepacc_qp_solve_batched.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [dp] * 8 + [dp, dp, dp, vp]
    lib.eepacc_synchronize.argtypes = [vp, vp]
    lib.eepacc_build_flags.restype = C.c_char_p
    _lib = lib
    return lib


ABI_SYMBOLS = ["eepacc_last_error", "eepacc_version", "eepacc_sizeof_settings", "eepacc_sizeof_vehicle", "eepacc_create", "eepacc_destroy", "eepacc_reset",
               "eepacc_ab_step", "eepacc_run_abmpc", "eepacc_fb_step", "eepacc_run_fbmpc",
               "eepacc_run_abmpc_host", "eepacc_run_fbmpc_host", "eepacc_bl_step", "eepacc_run_blmpc", "eepacc_run_blmpc_host",
               "eepacc_postprocess",
               "eepacc_last_iterations", "eepacc_qp_solve_batched", "eepacc_synchronize", "eepacc_build_flags"]


def _check(rc: int):
    if rc != 0:
        raise EepaccError(f"libeepacc error {rc}: {load_library().eepacc_last_error().decode()}")


class Engine:
    """Batched ABMPC engine bound to one GPU (one handle = one host thread / stream)."""

    def __init__(self, OPTsettings: Dict[str, Any], V: Dict[str, float], device: int = 0, max_batch: int = 4096):
        import torch
        if not torch.cuda.is_available():
            raise EepaccError("no GPU visible: the EEPACC engine has no CPU path")
        self.torch = torch
        self.lib = load_library()
        self.OPT = OPTsettings
        self.holder = SettingsHolder(OPTsettings)
        self.veh = make_vehicle(V)
        self.N = int(OPTsettings["N_hor"])
        self.device = torch.device("cuda", device)
        self.max_batch = int(max_batch)
        h = C.c_void_p()
        _check(self.lib.eepacc_create(C.byref(h), C.byref(self.holder.pod), C.byref(self.veh), device, max_batch))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.eepacc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _d(self, x, n):
        t = self.torch
        x = t.as_tensor(x, dtype=t.float64, device=self.device).contiguous()
        if x.numel() != n:
            raise ValueError(f"expected {n} values, got {x.numel()}")
        return x

    def reset(self):
        _check(self.lib.eepacc_reset(self.h))

    # B2 ------------------------------------------------------------------------------------
    def ab_step(self, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, want_pred: bool = True):
        t = self.torch
        B = int(t.as_tensor(s).numel())
        ins = [self._d(x, B) for x in (s, v, a_prev, t0, s_tv, v_tv, a_tv_prev)]
        out = t.empty((OUT_N, B), dtype=t.float64, device=self.device)
        sp = t.empty((self.N + 1, B), dtype=t.float64, device=self.device) if want_pred else None
        vp = t.empty((self.N + 1, B), dtype=t.float64, device=self.device) if want_pred else None
        status = t.empty((B,), dtype=t.int32, device=self.device)
        _check(self.lib.eepacc_ab_step(self.h, B, *[x.data_ptr() for x in ins], out.data_ptr(),
                                       sp.data_ptr() if want_pred else None,
                                       vp.data_ptr() if want_pred else None, status.data_ptr(), self._stream()))
        return out, sp, vp, status

    # B1 ------------------------------------------------------------------------------------
    def run_abmpc(self, s0, v0, a_minus1, s_tv, v_tv, resume: bool = False, out=None, by_name_bl: bool = False):
        """s_tv, v_tv: [n_steps, B] lead traces.  Returns traj [n_steps, OUT_N, B], status [n_steps, B].
        resume=True continues the simulation of the previous call (s_tv/v_tv hold the next rows).
        out=(traj, status): preallocated output tensors to write into."""
        t = self.torch
        if not resume:
            self.reset()
        s_tv = t.as_tensor(s_tv, dtype=t.float64, device=self.device).contiguous()
        v_tv = t.as_tensor(v_tv, dtype=t.float64, device=self.device).contiguous()
        n_steps, B = s_tv.shape
        ins = [self._d(x, B) for x in (s0, v0, a_minus1)]
        if out is not None:
            traj, status = out[0][:n_steps], out[1][:n_steps]
            assert traj.shape == (n_steps, OUT_N, B) and traj.is_contiguous() and status.is_contiguous()
        else:
            traj = t.empty((n_steps, OUT_N, B), dtype=t.float64, device=self.device)
            status = t.empty((n_steps, B), dtype=t.int32, device=self.device)
        f = self.lib.eepacc_run_blmpc if by_name_bl else self.lib.eepacc_run_abmpc
        _check(f(self.h, B, n_steps, *[x.data_ptr() for x in ins], s_tv.data_ptr(),
                 v_tv.data_ptr(), traj.data_ptr(), status.data_ptr(), self._stream()))
        return traj, status

    def run_blmpc(self, s0, v0, a_minus1, s_tv, v_tv, resume: bool = False, out=None):
        """eepacc_run_blmpc: run_abmpc on a handle created from settings.Settings_BL (refused on any other handle)."""
        return self.run_abmpc(s0, v0, a_minus1, s_tv, v_tv, resume=resume, out=out, by_name_bl=True)

    # FBMPC: same two operators (ABO/RunOpt_FBMPC.m:161-331) -----------------------------------
    def fb_step(self, s, v, v_prev, a_prev, Fm_prev, Fb_prev, t0, s_tv, v_tv, a_tv_prev, want_pred: bool = True):
        t = self.torch
        B = int(t.as_tensor(s).numel())
        ins = [self._d(x, B) for x in (s, v, v_prev, a_prev, Fm_prev, Fb_prev, t0, s_tv, v_tv, a_tv_prev)]
        out = t.empty((OUT_N, B), dtype=t.float64, device=self.device)
        sp = t.empty((self.N + 1, B), dtype=t.float64, device=self.device) if want_pred else None
        vp = t.empty((self.N + 1, B), dtype=t.float64, device=self.device) if want_pred else None
        status = t.empty((B,), dtype=t.int32, device=self.device)
        _check(self.lib.eepacc_fb_step(self.h, B, *[x.data_ptr() for x in ins], out.data_ptr(),
                                       sp.data_ptr() if want_pred else None,
                                       vp.data_ptr() if want_pred else None, status.data_ptr(), self._stream()))
        return out, sp, vp, status

    def run_fbmpc(self, s0, v0, a_minus1, s_tv, v_tv, resume: bool = False, out=None):
        """Closed-loop FBMPC; arguments and results as run_abmpc."""
        t = self.torch
        if not resume:
            self.reset()
        s_tv = t.as_tensor(s_tv, dtype=t.float64, device=self.device).contiguous()
        v_tv = t.as_tensor(v_tv, dtype=t.float64, device=self.device).contiguous()
        n_steps, B = s_tv.shape
        ins = [self._d(x, B) for x in (s0, v0, a_minus1)]
        if out is not None:
            traj, status = out[0][:n_steps], out[1][:n_steps]
            assert traj.shape == (n_steps, OUT_N, B) and traj.is_contiguous() and status.is_contiguous()
        else:
            traj = t.empty((n_steps, OUT_N, B), dtype=t.float64, device=self.device)
            status = t.empty((n_steps, B), dtype=t.int32, device=self.device)
        _check(self.lib.eepacc_run_fbmpc(self.h, B, n_steps, *[x.data_ptr() for x in ins], s_tv.data_ptr(),
                                         v_tv.data_ptr(), traj.data_ptr(), status.data_ptr(), self._stream()))
        return traj, status

    def postprocess(self, traj):
        t = self.torch
        n_steps, _, B = traj.shape
        outs = [t.empty((n_steps, B), dtype=t.float64, device=self.device) for _ in range(4)]
        _check(self.lib.eepacc_postprocess(self.h, B, n_steps, traj.data_ptr(), *[o.data_ptr() for o in outs],
                                           self._stream()))
        return outs   # rpm, Tm, P, E

    # B3 ------------------------------------------------------------------------------------
    def qp_solve_batched(self, H, g, A, lba=None, uba=None, lbx=None, ubx=None, x0=None):
        """sol = QPsolver('h',H,'g',g,'a',A,'lba',..,'uba',..,'lbx',..,'ubx',..) (ABO/RunOpt_ABMPC.m:252)
        for a batch.  H [B,nV,nV], g [B,nV], A [B,nC,nV] (row-major rows as in numpy; transposed
        here to the column-major layout of the C-ABI), bounds [B,nC] / [B,nV] or None.
        Returns x [B,nV], cost [B], status [B] as device tensors."""
        t = self.torch
        f64 = dict(dtype=t.float64, device=self.device)
        H = t.as_tensor(H, **f64).contiguous()
        B, nV = H.shape[0], H.shape[1]
        g = t.as_tensor(g, **f64).contiguous()
        A = t.as_tensor(A, **f64)
        nC = A.shape[1]
        A_cm = A.transpose(1, 2).contiguous()          # [B][nV][nC] = column-major nC x nV
        opt = lambda v, n: None if v is None else self._d(t.as_tensor(v, **f64).reshape(-1), B * n)
        lba, uba, lbx, ubx, x0 = opt(lba, nC), opt(uba, nC), opt(lbx, nV), opt(ubx, nV), opt(x0, nV)
        ptr = lambda v: None if v is None else v.data_ptr()
        x = t.empty((B, nV), **f64)
        cost = t.empty((B,), **f64)
        status = t.empty((B,), dtype=t.int32, device=self.device)
        _check(self.lib.eepacc_qp_solve_batched(self.h, B, nV, nC, H.data_ptr(), g.data_ptr(), A_cm.data_ptr(),
                                                ptr(lba), ptr(uba), ptr(lbx), ptr(ubx), ptr(x0), x.data_ptr(),
                                                cost.data_ptr(), status.data_ptr(), self._stream()))
        return x, cost, status

    def synchronize(self):
        """Wait for the engine's stream; raises if a closed-loop launch flagged a device-side failure."""
        _check(self.lib.eepacc_synchronize(self.h, self._stream()))

    def _run_host(self, fn, s0, v0, a_minus1, s_tv, v_tv):
        s_tv = np.ascontiguousarray(s_tv, dtype=np.float64); v_tv = np.ascontiguousarray(v_tv, dtype=np.float64)
        n_steps, B = s_tv.shape
        ins = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1)) for x in (s0, v0, a_minus1)]
        assert all(x.size == B for x in ins)
        traj = np.empty((n_steps, OUT_N, B)); status = np.empty((n_steps, B), dtype=np.int32)
        _check(fn(self.h, B, n_steps, *[as_dptr(x) for x in ins], as_dptr(s_tv), as_dptr(v_tv), as_dptr(traj),
                  status.ctypes.data_as(C.POINTER(C.c_int32))))
        return traj, status

    def run_abmpc_host(self, s0, v0, a_minus1, s_tv, v_tv):
        """eepacc_run_abmpc_host: host (numpy) buffers in and out -- the entry a MEX gateway calls."""
        return self._run_host(self.lib.eepacc_run_abmpc_host, s0, v0, a_minus1, s_tv, v_tv)

    def run_fbmpc_host(self, s0, v0, a_minus1, s_tv, v_tv):
        return self._run_host(self.lib.eepacc_run_fbmpc_host, s0, v0, a_minus1, s_tv, v_tv)

    def last_iterations(self, B):
        it = np.zeros(B, dtype=np.int32)
        _check(self.lib.eepacc_last_iterations(self.h, B, it.ctypes.data_as(C.POINTER(C.c_int32))))
        return it


def RunOpt_ABMPC(OPTsettings: Dict[str, Any], V: Optional[Dict[str, float]] = None, device: int = 0) -> Dict[str, Any]:
    """optSol = RunOpt_ABMPC(OPTsettings)  -- ABO/RunOpt_ABMPC.m:1, single ego vehicle.

    Same inputs (fields of OPTsettings incl. s_tv, v_tv, t_sim, s_init, v_init, a_minus1) and
    the same optSol fields (:354-404) except the wall-clock vectors tLoop/tSolve."""
    from .settings import SetVehicleParameters
    if V is None:
        V = SetVehicleParameters(OPTsettings.get("tree", "ABO"))
    eng = Engine(OPTsettings, V, device=device, max_batch=1)
    t = eng.torch
    Ts = float(OPTsettings["Tvec"][0])
    n_steps = int(round(OPTsettings["t_sim"] / Ts)) + 1
    s_tv = np.asarray(OPTsettings["s_tv"], dtype=np.float64).reshape(-1)[:n_steps].reshape(n_steps, 1)
    v_tv = np.asarray(OPTsettings["v_tv"], dtype=np.float64).reshape(-1)[:n_steps].reshape(n_steps, 1)
    traj, status = eng.run_abmpc([OPTsettings["s_init"]], [OPTsettings["v_init"]], [OPTsettings["a_minus1"]], s_tv, v_tv)
    rpm, Tm, P, E = eng.postprocess(traj)
    t.cuda.synchronize()
    tr = traj.cpu().numpy()[:, :, 0]
    sol: Dict[str, Any] = {}
    for name, key in (("s", "s_opt"), ("v", "v_opt"), ("Fm", "Fm_opt"), ("Fb", "Fb_opt"), ("xi_v", "xi_v_opt"),
                      ("xi_h", "xi_h_opt"), ("xi_s", "xi_s_opt"), ("xi_f", "xi_f_opt"), ("a", "a_opt"),
                      ("DistHor", "DistHor"), ("cost", "cost")):
        sol[key] = tr[:, OUT[name]].copy()
    sol["exitMessage"] = status.cpu().numpy()[:, 0].astype(np.float64)
    sol["rpm_opt"] = rpm.cpu().numpy()[:, 0]; sol["Tm_opt"] = Tm.cpu().numpy()[:, 0]
    sol["P_opt"] = P.cpu().numpy()[:, 0]; sol["E_opt"] = E.cpu().numpy()[:, 0]
    sol["j_opt"] = np.diff(sol["a_opt"]) / Ts
    W = np.asarray(OPTsettings["W_AB"]).ravel()      # cost_* use W(1..5) as the reference does (:383-388)
    N_sim = n_steps - 1
    for nm, w, arr in (("cost_a", W[0], sol["a_opt"] ** 2), ("cost_j", W[1], sol["j_opt"] ** 2),
                       ("cost_xi_v", W[2], sol["xi_v_opt"]), ("cost_xi_h", W[3], sol["xi_h_opt"]),
                       ("cost_xi_s", W[4], sol["xi_s_opt"]), ("cost_xi_f", W[4], sol["xi_f_opt"])):
        sol[nm] = w * np.cumsum(arr)[:N_sim]
    return sol


def RunOpt_BLMPC(OPTsettings: Dict[str, Any], V: Optional[Dict[str, float]] = None, device: int = 0) -> Dict[str, Any]:
    """optSol = RunOpt_BLMPC(OPTsettings)  -- ABO/RunOpt_BLMPC.m:1, the baseline controller, single ego vehicle.

    Takes the same OPTsettings as the other controllers (the BL_* fields and W_BL select horizon, estimator and
    weights, settings.Settings_BL) and returns the optSol fields of :318-345."""
    from .settings import SetVehicleParameters, Settings_BL
    if V is None:
        V = SetVehicleParameters(OPTsettings.get("tree", "ABO"))
    BL = Settings_BL(OPTsettings)
    eng = Engine(BL, V, device=device, max_batch=1)
    t = eng.torch
    Ts = float(BL["Tvec"][0])
    n_steps = int(round(OPTsettings["t_sim"] / Ts)) + 1
    s_tv = np.asarray(OPTsettings["s_tv"], dtype=np.float64).reshape(-1)[:n_steps].reshape(n_steps, 1)
    v_tv = np.asarray(OPTsettings["v_tv"], dtype=np.float64).reshape(-1)[:n_steps].reshape(n_steps, 1)
    traj, status = eng.run_blmpc([OPTsettings["s_init"]], [OPTsettings["v_init"]], [OPTsettings["a_minus1"]], s_tv, v_tv)
    rpm, Tm, P, E = eng.postprocess(traj)
    t.cuda.synchronize()
    tr = traj.cpu().numpy()[:, :, 0]
    sol: Dict[str, Any] = {}
    for name, key in (("s", "s_opt"), ("v", "v_opt"), ("Fm", "Fm_opt"), ("Fb", "Fb_opt"), ("xi_f", "xi_f_opt"),
                      ("a", "a_opt"), ("cost", "cost")):
        sol[key] = tr[:, OUT[name]].copy()
    sol["exitMessage"] = status.cpu().numpy()[:, 0].astype(np.float64)
    sol["rpm_opt"] = rpm.cpu().numpy()[:, 0]; sol["Tm_opt"] = Tm.cpu().numpy()[:, 0]
    sol["P_opt"] = P.cpu().numpy()[:, 0]; sol["E_opt"] = E.cpu().numpy()[:, 0]
    sol["j_opt"] = np.diff(sol["a_opt"]) / Ts
    return sol


def RunOpt_FBMPC(OPTsettings: Dict[str, Any], V: Optional[Dict[str, float]] = None, device: int = 0) -> Dict[str, Any]:
    """optSol = RunOpt_FBMPC(OPTsettings)  -- ABO/RunOpt_FBMPC.m:1, single ego vehicle.

    Same inputs and optSol fields (:345-397) except tLoop/tSolve and the final-step H, G."""
    from .settings import SetVehicleParameters
    if V is None:
        V = SetVehicleParameters(OPTsettings.get("tree", "ABO"))
    eng = Engine(OPTsettings, V, device=device, max_batch=1)
    t = eng.torch
    Ts = float(OPTsettings["Tvec"][0])
    n_steps = int(round(OPTsettings["t_sim"] / Ts)) + 1
    s_tv = np.asarray(OPTsettings["s_tv"], dtype=np.float64).reshape(-1)[:n_steps].reshape(n_steps, 1)
    v_tv = np.asarray(OPTsettings["v_tv"], dtype=np.float64).reshape(-1)[:n_steps].reshape(n_steps, 1)
    traj, status = eng.run_fbmpc([OPTsettings["s_init"]], [OPTsettings["v_init"]], [OPTsettings["a_minus1"]], s_tv, v_tv)
    rpm, Tm, P, E = eng.postprocess(traj)
    t.cuda.synchronize()
    tr = traj.cpu().numpy()[:, :, 0]
    sol: Dict[str, Any] = {}
    for name, key in (("s", "s_opt"), ("v", "v_opt"), ("Fm", "Fm_opt"), ("Fb", "Fb_opt"), ("xi_v", "xi_v_opt"),
                      ("xi_h", "xi_h_opt"), ("xi_s", "xi_s_opt"), ("xi_f", "xi_f_opt"), ("a", "a_opt"),
                      ("DistHor", "DistHor"), ("cost", "cost")):
        sol[key] = tr[:, OUT[name]].copy()
    sol["exitMessage"] = status.cpu().numpy()[:, 0].astype(np.float64)
    sol["rpm_opt"] = rpm.cpu().numpy()[:, 0]; sol["Tm_opt"] = Tm.cpu().numpy()[:, 0]
    sol["P_opt"] = P.cpu().numpy()[:, 0]; sol["E_opt"] = E.cpu().numpy()[:, 0]
    sol["j_opt"] = np.diff(sol["a_opt"]) / Ts
    W = np.asarray(OPTsettings["W_FB"]).ravel()      # :373-390
    N_sim = n_steps - 1
    for nm, w, arr in (("cost_P", W[0], sol["P_opt"] ** 2), ("cost_a", W[1], sol["a_opt"] ** 2),
                       ("cost_j", W[2], sol["j_opt"] ** 2), ("cost_xi_v", W[3], sol["xi_v_opt"]),
                       ("cost_xi_h", W[4], sol["xi_h_opt"]), ("cost_xi_s", W[5], sol["xi_s_opt"]),
                       ("cost_xi_f", W[6], sol["xi_f_opt"])):
        sol[nm] = w * np.cumsum(arr)[:N_sim]
    return sol
