"""The MEX gateways executed end to end without MATLAB: each mex/RunOpt_*.c is compiled with a small functional stand-in
of the MEX runtime (tests/mexstub/mex_mock.c: mxArray, struct fields, mexCallMATLAB("SetVehicleParameters"),
mexErrMsgIdAndTxt) and libeepacc, run as a process on an OPTsettings file -- one struct in, one struct out, exactly the
call `optSol = RunOpt_X(OPTsettings)` of ABO/Main.m:97,106,115 -- and its optSol is compared field by field with the
Python mirror of the same entry point (and through it with the reference's saved solutions).  Needs the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_case, load_golden

pytestmark = pytest.mark.gpu
PKG = os.path.join(ROOT, "eepacc_mpc_casadi_matlab_amd")


def _write_struct(path, OPT, V):
    def entry(name, val):
        a = np.asarray(val, dtype=np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(1, -1)                      # row vector, as Settings.m writes them
        m, n = a.shape
        vals = " ".join(("inf" if v == np.inf else ("-inf" if v == -np.inf else repr(float(v)))) for v in a.ravel(order="F"))
        return "%s %d %d %s" % (name, m, n, vals)
    lines = []
    for k, v in OPT.items():
        if isinstance(v, (bool, int, float, np.integer, np.floating)) or (isinstance(v, np.ndarray) and v.dtype.kind in "fiub"):
            lines.append(entry(k, v))
        elif isinstance(v, (list, tuple)) and len(v) and all(isinstance(x, (int, float)) for x in v):
            lines.append(entry(k, v))
    for k, v in V.items():
        lines.append(entry("V." + k, v))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def _read_struct(path):
    out = {}
    for line in open(path):
        parts = line.split()
        if parts[1] == "str":
            out[parts[0]] = " ".join(parts[2:])
        else:
            m, n = int(parts[1]), int(parts[2])
            out[parts[0]] = np.array([float(x) for x in parts[3:3 + m * n]]).reshape((m, n), order="F")
    return out


def _run_gateway(name, OPT, V, tmp_path):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-std=c99", "-O1", os.path.join(ROOT, "mex", name + ".c"), os.path.join(ROOT, "tests", "mexstub", "mex_mock.c"),
           "-I", os.path.join(ROOT, "tests", "mexstub"), "-I", os.path.join(ROOT, "include"), "-L", PKG, "-leepacc", "-lm",
           "-Wl,-rpath," + PKG, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fin, fout = str(tmp_path / (name + "_in.txt")), str(tmp_path / (name + "_out.txt"))
    _write_struct(fin, OPT, V)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return _read_struct(fout)


def _case(tree, t_sim):
    OPT, V, s_tv, v_tv = make_case(tree, 20)
    OPT = dict(OPT); OPT["t_sim"] = t_sim
    n = int(round(t_sim / 0.5)) + 1
    OPT["s_tv"] = s_tv[:n].copy(); OPT["v_tv"] = v_tv[:n].copy()
    return OPT, V


@pytest.mark.parametrize("tree", ["ABO", "ORIG"])
def test_runopt_abmpc_gateway(tree, tmp_path):
    from eepacc_mpc_casadi_matlab_amd.engine import RunOpt_ABMPC
    OPT, V = _case(tree, 100.0)
    S = _run_gateway("RunOpt_ABMPC", OPT, V, tmp_path)
    P = RunOpt_ABMPC(OPT, V)
    G = load_golden(f"{tree.lower()}_abmpc")
    n = 201
    for k in ("s_opt", "v_opt", "Fm_opt", "Fb_opt", "a_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt", "DistHor"):
        assert S[k].shape == (n, 1), k
        np.testing.assert_array_equal(S[k].ravel(), np.asarray(P[k]).ravel(), err_msg=k)           # same kernels, same bits
        assert np.abs(S[k].ravel() - G[k][:n]).max() < (1e-6 if k in ("Fm_opt", "Fb_opt") else 1e-8), k
    for k in ("P_opt", "E_opt", "Tm_opt", "rpm_opt", "j_opt"):
        ref = np.asarray(G[k], dtype=np.float64).ravel()[:S[k].size]
        assert np.abs(S[k].ravel() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k
    # cumulative cost series with the reference's index shift (RunOpt_ABMPC.m:381-388: W(1..5), W(5) twice): a slack that is
    # 1e-12 where the saved one is exactly zero is multiplied by weights up to 9e7, so the tolerance carries the weight
    W = np.asarray(OPT["W_AB"], dtype=np.float64).ravel()
    for k, w in zip(("cost_a", "cost_j", "cost_xi_v", "cost_xi_h", "cost_xi_s", "cost_xi_f"), (W[0], W[1], W[2], W[3], W[4], W[4])):
        ref = np.asarray(G[k], dtype=np.float64).ravel()[:S[k].size]
        assert S[k].size == n - 1
        assert np.abs(S[k].ravel() - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()) + 1e-9 * w * n, k
    assert S["exitMessage"].shape == (1, n) and S["exitMessage"].sum() == 0
    assert S["H"].size == 0 and S["G"].size == 0 and S["tLoop"].shape == (n, 1)


def test_runopt_fbmpc_and_blmpc_gateways(tmp_path):
    from eepacc_mpc_casadi_matlab_amd.engine import RunOpt_FBMPC, RunOpt_BLMPC
    OPT, V = _case("ABO", 60.0)
    S = _run_gateway("RunOpt_FBMPC", OPT, V, tmp_path)
    P = RunOpt_FBMPC(OPT, V)
    G = load_golden("abo_fbmpc")
    n = 121
    for k in ("s_opt", "v_opt", "Fm_opt", "Fb_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt"):
        np.testing.assert_array_equal(S[k].ravel(), np.asarray(P[k]).ravel(), err_msg=k)
    for k in ("s_opt", "v_opt", "xi_v_opt", "xi_h_opt"):
        assert np.abs(S[k].ravel() - G[k][:n]).max() < 1e-8, k
    assert "cost_P" in S and S["exitMessage"].sum() == 0
    S = _run_gateway("RunOpt_BLMPC", OPT, V, tmp_path)
    P = RunOpt_BLMPC(OPT, V)
    assert "xi_v_opt" not in S and "DistHor" not in S                 # not part of RunOpt_BLMPC's optSol (RunOpt_BLMPC.m:318-345)
    for k in ("s_opt", "v_opt", "Fm_opt", "Fb_opt", "a_opt"):
        np.testing.assert_array_equal(S[k].ravel(), np.asarray(P[k]).ravel(), err_msg=k)
    Gb = load_golden("abo_blmpc")
    assert np.abs(S["s_opt"].ravel()[:40] - Gb["s_opt"][:40]).max() < 1e-5           # up to the saved solution's degenerate step


def test_runopt_nlp_gateway(tmp_path):
    """NLPsol = RunOpt_NLP(OPTsettings) through mex/RunOpt_NLP.c: 60 s of the reference scenario from a cold start, against
    the Python mirror (same entry point eepacc_run_nlp_host, same start set), field by field; a malformed struct throws."""
    from eepacc_mpc_casadi_matlab_amd.nlp import RunOpt_NLP
    OPT, V = _case("ABO", 60.0)
    OPT.update(shootingMethod=1, discretizationMethod=0, useFifthOrderFit_NLP=True, NLPmaxIter=300)
    S = _run_gateway("RunOpt_NLP", OPT, V, tmp_path)
    P = RunOpt_NLP(OPT, V)
    assert S["exitMessage"] == P["exitMessage"] == "Solve_Succeeded"
    N = 120
    for k in ("s_opt", "v_opt", "theta_opt", "j_opt"):
        assert S[k].shape == (N + 1, 1), k
    for k in ("s_opt", "v_opt", "theta_opt", "j_opt", "Fm_opt", "Fb_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt"):
        np.testing.assert_array_equal(S[k].ravel(), np.asarray(P[k]).ravel(), err_msg=k)           # same solver, same bits
    for k in ("P_opt", "E_opt", "a_opt", "Tm_opt", "rpm_opt", "cost_P", "cost_a", "cost_j", "cost_xi_v", "cost_xi_h", "cost_xi_s", "cost_xi_f"):
        ref = np.asarray(P[k], dtype=np.float64).ravel()
        assert S[k].size == N and np.abs(S[k].ravel() - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), k
    np.testing.assert_allclose(S["s_velInc"].ravel(), P["s_velInc"], rtol=0, atol=1e-12)
    assert float(S["J"][0, 0]) == P["J"] and int(S["iterations"][0, 0]) == P["iterations"]
    # malformed input throws instead of running
    bad = dict(OPT); bad["W_NLP"] = np.ones(5)
    exe = str(tmp_path / "RunOpt_NLP")
    fin = str(tmp_path / "bad.txt")
    _write_struct(fin, bad, V)
    r = subprocess.run([exe, fin, str(tmp_path / "bad_out.txt")], capture_output=True, text=True)
    assert r.returncode == 2 and "W_NLP must have 7 entries" in r.stderr
