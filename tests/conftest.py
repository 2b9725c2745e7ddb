import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lead_trace():
    return np.load(os.path.join(GOLDEN, "lead_TO01_EAD.npz"))


# Saved ABMPC solutions of the ABO tree written with other weight sets than the checked-in Settings.m (its commented
# alternatives, ABO/Settings.m:48-64: w_FC % 50, w_c % 0.1, w_v % 8e5, w_h % 1e4, w_f % 1e10, W_AB = 1e-3*W).  The
# reference does not store OPTsettings with a solution; the weights below are recovered from the files themselves:
# W(1..4) from the cost_* series (cost_a = W(1) cumsum(a^2), ... -- the index shift of RunOpt_ABMPC.m:381-388), w_h
# from the xi_h diagonal of the saved H (2 w_h), w_FC p01 F2 from H(1,1); w_s, w_f from the comments (w_f = 1e-3 * 1e10,
# w_s = 9 w_f) -- EFFMAP has xi_s, xi_f > 0 and matches with them.  Vehicle constants: the checked-in ones.
GOLDEN_AB_VARIANTS = {
    "abo_abmpc_effmap":  [1000.0, 30.0, 1000.0, 800.0, 0.1, 9e7, 1e7],
    "abo_abmpc_fcopt":   [0.05, 30.0, 1000.0, 800.0, 10.0, 9e7, 1e7],
    "abo_abmpc_nofcopt": [0.0, 3.0, 100.0, 800.0, 10.0, 9e7, 1e7],
}


# savedABMPCsolICEMAP.mat: the EFFMAP weights (same cost_* ratios, xi_h diagonal 0.2) with the ICE-map fuel term of
# CreateQP_AB.m:154-159 (commented in the checked-in file): k10, k01 of SetVehicleParameters.m:44-46, gear ratio per
# horizon stage from LUTgearshift.m -- the saved H's per-stage v^2 coefficient at standstill (first gear) is
# 2*1000*k01*F2*R_w/tau_fd/4.714/eta_drive = 0.65144.
GOLDEN_AB_ICEMAP = dict(W_AB=[1000.0, 30.0, 1000.0, 800.0, 0.1, 9e7, 1e7], fuel_map="ICE")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def make_case(tree="ABO", N_hor=20, lead=None, **overrides):
    """Settings + vehicle + lead trace exactly as ABO/Main.m:44-89 sets them up."""
    from eepacc_mpc_casadi_matlab_amd.settings import Settings, SetVehicleParameters, Run_DrivingCycle
    OPT = Settings(tree=tree, N_hor=N_hor)
    OPT.update(overrides)
    V = SetVehicleParameters(tree)
    if lead is None:
        lead = np.load(os.path.join(GOLDEN, "lead_TO01_EAD.npz"))
    s_tv, v_tv = Run_DrivingCycle(OPT, V_TO_resampled=lead["V_TO_2Hz"])
    s_tv = s_tv - OPT["TVlength"]          # ABO/Main.m:88
    return OPT, V, s_tv, v_tv


def golden_step_inputs(G, s_tv, v_tv, k, Ts=0.5):
    """Open-loop inputs of MPC step k reconstructed from a golden trajectory
    (measurement block ABO/RunOpt_ABMPC.m:159-191)."""
    s, v = float(G["s_opt"][k]), float(G["v_opt"][k])
    if k == 0:
        return dict(s=s, v=v, a_prev=0.0, t0=0.0, s_tv=float(s_tv[0]), v_tv=0.0, a_tv_prev=0.0)
    a_prev = (v - float(G["v_opt"][k - 1])) / Ts
    vtv = float(v_tv[k])
    vtv_prev = float(v_tv[k - 1]) if k > 1 else 0.0
    return dict(s=s, v=v, a_prev=a_prev, t0=k * Ts, s_tv=float(s_tv[k]), v_tv=vtv,
                a_tv_prev=(vtv - vtv_prev) / Ts)
