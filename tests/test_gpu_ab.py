"""Parity of the HIP ABMPC path (through the C-ABI) with the CPU oracle and the reference goldens.

Tolerances (fp64, stated per SURVEY.md section 7.1 / BASELINE.md section 3): slacks, speeds,
accelerations 1e-9 (absolute, SI units); positions 1e-8 m; forces 1e-6 N; QP cost 1e-8 relative.
"""
import numpy as np
import pytest

from conftest import make_case, load_golden, golden_step_inputs, GOLDEN_AB_VARIANTS, GOLDEN_AB_ICEMAP
from eepacc_mpc_casadi_matlab_amd._abi import OUT, OUT_N
from eepacc_mpc_casadi_matlab_amd.scenarios import make_s1, make_s2

pytestmark = pytest.mark.gpu

TOL = dict(s=1e-8, v=1e-9, Fm=1e-6, Fb=1e-6, a=1e-9, xi_v=1e-9, xi_h=1e-9, xi_s=1e-9, xi_f=1e-9,
           DistHor=1e-8, a_qp=1e-9)


def _engine(OPT, V, max_batch=4096):
    from eepacc_mpc_casadi_matlab_amd.engine import Engine
    return Engine(OPT, V, device=0, max_batch=max_batch)


def _cols(inps):
    return {n: np.array([d[n] for d in inps]) for n in ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")}


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("tree", ["ABO", "ORIG"])
def test_open_loop_all_golden_steps(tree, torch_mod):
    """Every one of the 871 saved MPC steps as an independent cold-started QP."""
    OPT, V, s_tv, v_tv = make_case(tree, 20)
    G = load_golden(f"{tree.lower()}_abmpc")
    eng = _engine(OPT, V)
    c = _cols([golden_step_inputs(G, s_tv, v_tv, k) for k in range(871)])
    out, sp, vp, status = eng.ab_step(**c)
    o = out.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    for n in ("xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a", "DistHor"):
        g = G[n if n == "DistHor" else n + "_opt"]
        assert np.abs(o[OUT[n]] - g).max() < TOL[n], n
    np.testing.assert_array_equal(o[OUT["s"]], G["s_opt"])


@pytest.mark.parametrize("tree", ["ABO", "ORIG"])
def test_closed_loop_golden_trajectory(tree, torch_mod):
    OPT, V, s_tv, v_tv = make_case(tree, 20)
    G = load_golden(f"{tree.lower()}_abmpc")
    eng = _engine(OPT, V, 8)
    B = 5                                   # odd: exercises a partially filled block
    stv = np.repeat(s_tv[:871, None], B, 1); vtv = np.repeat(v_tv[:871, None], B, 1)
    traj, status = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
    tr = traj.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    for n in ("s", "v", "xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a", "DistHor"):
        g = G[n if n == "DistHor" else n + "_opt"]
        assert np.abs(tr[:, OUT[n], 0] - g).max() < TOL[n], n
    assert np.abs(tr - tr[:, :, :1]).max() == 0.0          # identical instances, identical results
    rpm, Tm, P, E = [x.cpu().numpy()[:, 0] for x in eng.postprocess(traj)]
    for a, b in ((rpm, "rpm_opt"), (Tm, "Tm_opt"), (P, "P_opt"), (E, "E_opt")):
        assert (np.abs(a - G[b]) / np.maximum(1.0, np.abs(G[b]))).max() < 1e-9, b
    # chunked run (resume) is bit-identical to the single launch
    t1, s1 = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv[:300], vtv[:300])
    t2, s2 = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv[300:], vtv[300:], resume=True)
    both = np.concatenate([t1.cpu().numpy(), t2.cpu().numpy()], 0)
    assert np.array_equal(both, tr)


@pytest.mark.parametrize("name", sorted(GOLDEN_AB_VARIANTS))
def test_golden_weight_variants(name, torch_mod):
    """The three further saved ABMPC solutions of the ABO tree (other weight sets, conftest.GOLDEN_AB_VARIANTS): all
    871 steps as cold QPs and as one closed loop."""
    OPT, V, s_tv, v_tv = make_case("ABO", 20)
    OPT = dict(OPT); OPT["W_AB"] = np.array(GOLDEN_AB_VARIANTS[name])
    G = load_golden(name)
    eng = _engine(OPT, V)
    c = _cols([golden_step_inputs(G, s_tv, v_tv, k) for k in range(871)])
    out, sp, vp, status = eng.ab_step(**c)
    o = out.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    tol = dict(TOL, Fm=1e-5, Fb=1e-5)                # weights up to 1e7: forces to 1e-5 N
    for n in ("xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a", "DistHor"):
        g = G[n if n == "DistHor" else n + "_opt"]
        assert np.abs(o[OUT[n]] - g).max() < tol[n], n
    B = 2
    stv = np.repeat(s_tv[:871, None], B, 1); vtv = np.repeat(v_tv[:871, None], B, 1)
    traj, status = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
    eng.synchronize()
    tr = traj.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    for n in ("s", "v", "xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a"):
        assert np.abs(tr[:, OUT[n], 0] - G[n + "_opt"]).max() < 10 * tol[n], n
    E = eng.postprocess(traj)[3].cpu().numpy()[:, 0]
    assert abs(E[-1] - G["E_opt"][-1]) < 1e-9 * abs(G["E_opt"][-1])


def test_golden_icemap_step_varying_hessian(torch_mod, lead_trace):
    """savedABMPCsolICEMAP.mat: ABMPC with the ICE-map fuel term (CreateQP_AB.m:154-159).  The gear ratio of every horizon
    stage follows the estimated speed (LUTgearshift.m), so the condensed Hessian changes from step to step: the kernel
    variant `ice` builds it from closed forms and inverts it in LDS every step.  All 871 saved steps as cold QPs and as one
    closed loop against the golden; S2 scenarios at N = 30 against the oracle (gears 1-5 along the horizon)."""
    from oracle import Oracle
    OPT, V, s_tv, v_tv = make_case("ABO", 20)
    OPT = dict(OPT); OPT["W_AB"] = np.array(GOLDEN_AB_ICEMAP["W_AB"]); OPT["fuel_map"] = GOLDEN_AB_ICEMAP["fuel_map"]
    G = load_golden("abo_abmpc_icemap")
    eng = _engine(OPT, V)
    c = _cols([golden_step_inputs(G, s_tv, v_tv, k) for k in range(871)])
    out, sp, vp, status = eng.ab_step(**c)
    o = out.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    tol = dict(TOL, Fm=1e-5, Fb=1e-5)                # weights up to 1e7: forces to 1e-5 N (as the other weight variants)
    for n in ("xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a", "DistHor"):
        g = G[n if n == "DistHor" else n + "_opt"]
        assert np.abs(o[OUT[n]] - g).max() < tol[n], n
    B = 3
    stv = np.repeat(s_tv[:871, None], B, 1); vtv = np.repeat(v_tv[:871, None], B, 1)
    traj, status = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
    eng.synchronize()
    tr = traj.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    assert np.abs(tr - tr[:, :, :1]).max() == 0.0
    for n in ("s", "v", "xi_v", "xi_h", "xi_s", "xi_f", "Fm", "Fb", "a"):
        assert np.abs(tr[:, OUT[n], 0] - G[n + "_opt"]).max() < 10 * tol[n], n
    E = eng.postprocess(traj)[3].cpu().numpy()[:, 0]
    assert abs(E[-1] - G["E_opt"][-1]) < 1e-9 * abs(G["E_opt"][-1])
    # N = 30, S2 scenarios, against the oracle's closed loop
    OPT30, V30, _, _ = make_case("ABO", 30)
    OPT30 = dict(OPT30); OPT30["W_AB"] = np.array(GOLDEN_AB_ICEMAP["W_AB"]); OPT30["fuel_map"] = "ICE"
    Bs, n_steps = 6, 100
    sc = make_s2(Bs, n_steps, lead_trace["V_TO_2Hz"])
    eng30 = _engine(OPT30, V30, 8)
    traj, status = eng30.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    tr = traj.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    orc = Oracle(OPT30, V30)
    gears = set()
    for i in range(Bs):
        ref, st, _ = orc.run("ab", n_steps, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
        assert st.sum() == 0
        for n in ("s", "v", "Fm", "Fb", "a", "xi_v", "xi_h", "xi_s", "xi_f"):
            assert np.abs(tr[:, OUT[n], i] - ref[:, OUT[n]]).max() < 10 * tol[n], (i, n)
        gears |= {orc.lib_lut(v) for v in ref[:, OUT["v"]]}
    assert len(gears) >= 3, gears


@pytest.mark.parametrize("tree,N,B", [("ABO", 20, 192), ("ABO", 30, 96), ("ORIG", 30, 48), ("ABO", 60, 12)])
def test_open_loop_seeded_batch_vs_oracle(tree, N, B, torch_mod):
    """S1 inputs (perturbed golden states): every output incl. predicted trajectories and cost."""
    from oracle import Oracle
    OPT, V, s_tv, v_tv = make_case(tree, N)
    G = load_golden(f"{tree.lower()}_abmpc")
    s1 = make_s1(B, G, s_tv, v_tv)
    eng = _engine(OPT, V)
    args = {k: s1[k] for k in ("s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev")}
    out, sp, vp, status = eng.ab_step(**args)
    o = out.cpu().numpy(); sp = sp.cpu().numpy(); vp = vp.cpu().numpy(); st = status.cpu().numpy()
    orc = Oracle(OPT, V)
    # the ORIG weights reach 1e7 (ORIG/Settings.m:48-62): its QPs are ~1e4 times worse conditioned,
    # forces are compared to 1e-5 N there (relative 1e-8 of the force range)
    tol = dict(TOL)
    if tree == "ORIG":
        tol.update(Fm=1e-5, Fb=1e-5)
    for i in range(B):
        r = orc.ab_step(**{k: float(v[i]) for k, v in args.items()})
        assert r["status"] == st[i] == 0, i
        for n, t in tol.items():
            assert abs(o[OUT[n], i] - r["out"][OUT[n]]) < t, (i, n)
        ctol = 1e-6 if tree == "ORIG" else 1e-8
        assert abs(o[OUT["cost"], i] - r["out"][OUT["cost"]]) < ctol * (1 + abs(r["out"][OUT["cost"]])), i
        assert np.abs(sp[:, i] - r["s_pred"]).max() < (1e-6 if tree == "ORIG" else 1e-8)
        assert np.abs(vp[:, i] - r["v_pred"]).max() < (1e-7 if tree == "ORIG" else 1e-9)


def test_closed_loop_s2_vs_oracle(torch_mod, lead_trace):
    """N=30 closed loop on synthetic S2 scenarios against the oracle's closed loop."""
    from oracle import Oracle
    OPT, V, _, _ = make_case("ABO", 30)
    B, n_steps = 6, 120
    sc = make_s2(B, n_steps, lead_trace["V_TO_2Hz"])
    eng = _engine(OPT, V, 8)
    traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    tr = traj.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    orc = Oracle(OPT, V)
    for i in range(B):
        ref, st, _ = orc.run("ab", n_steps, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
        assert st.sum() == 0
        for n in ("s", "v", "Fm", "Fb", "a", "xi_v", "xi_h", "xi_s", "xi_f"):
            assert np.abs(tr[:, OUT[n], i] - ref[:, OUT[n]]).max() < 10 * TOL[n], (i, n)


@pytest.mark.parametrize("N,B", [(30, 4096), (60, 8192)])
def test_full_size_batch_properties(N, B, torch_mod, lead_trace):
    """BASELINE config 2 size (N=30, batch 4096) and config 4's share of one GPU (N=60, 65536 / 8 = 8192 instances):
    size-independent properties."""
    torch = torch_mod
    OPT, V, _, _ = make_case("ABO", N)
    n_steps = 40
    sc = make_s2(B, n_steps, lead_trace["V_TO_2Hz"])
    eng = _engine(OPT, V, B)
    traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    tr = traj.cpu().numpy(); st = status.cpu().numpy()
    assert st.sum() == 0
    assert np.isfinite(tr).all()
    # (1) determinism
    traj2, _ = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    assert np.array_equal(traj2.cpu().numpy(), tr)
    # (2) instances are independent: a permuted batch gives the permuted result
    perm = np.random.default_rng(0).permutation(B)
    trp, _ = eng.run_abmpc(sc["s0"][perm], sc["v0"][perm], sc["a_minus1"][perm],
                           np.ascontiguousarray(sc["s_tv"][:, perm]), np.ascontiguousarray(sc["v_tv"][:, perm]))
    assert np.array_equal(trp.cpu().numpy(), tr[:, :, perm])
    # (3) constraints the QP enforces hold on the applied trajectory: slacks >= 0, v >= 0
    for n in ("xi_v", "xi_h", "xi_s", "xi_f"):
        assert tr[:, OUT[n]].min() > -1e-9
    assert tr[:, OUT["v"]].min() > -1e-9
    # (4) the per-step operator reproduces step k of the closed loop from its inputs
    k = 17
    Ts = 0.5
    s, v = tr[k, OUT["s"]], tr[k, OUT["v"]]
    a_prev = (v - tr[k - 1, OUT["v"]]) / Ts
    vtv = sc["v_tv"][k]; vtvp = sc["v_tv"][k - 1]
    eng.reset()
    out, _, _, st1 = eng.ab_step(s, v, a_prev, np.full(B, k * Ts), sc["s_tv"][k], vtv, (vtv - vtvp) / Ts)
    o = out.cpu().numpy()
    assert st1.cpu().numpy().sum() == 0
    for n in ("Fm", "Fb", "a", "xi_v", "xi_h", "xi_s", "xi_f"):
        assert np.abs(o[OUT[n]] - tr[k, OUT[n]]).max() < 10 * TOL[n], n


def test_edge_cases(torch_mod, lead_trace):
    OPT, V, s_tv, v_tv = make_case("ABO", 20)
    eng = _engine(OPT, V, 16)
    # B = 1, one step, standstill behind a standing lead: nothing to do but wait
    out, sp, vp, st = eng.ab_step([0.0], [0.0], [0.0], [0.0], [6.0], [0.0], [0.0])
    o = out.cpu().numpy()[:, 0]
    assert st.cpu().numpy()[0] == 0
    assert abs(o[OUT["xi_v"]] - 60 / 3.6) < 1e-9
    # lead far away (no headway rows active) and lead closer than the minimum gap (safety slack)
    out, _, _, st = eng.ab_step([0.0, 0.0], [10.0, 10.0], [0.0, 0.0], [0.0, 0.0], [1e4, 1.0], [10.0, 0.0], [0.0, 0.0])
    o = out.cpu().numpy()
    assert st.cpu().numpy().sum() == 0
    assert o[OUT["xi_h"], 0] == 0.0 and o[OUT["xi_s"], 0] == 0.0
    assert o[OUT["xi_s"], 1] > 0.0 and o[OUT["xi_h"], 1] > 0.0
    # infeasible hard row (v0 above v_max): status 1, outputs still finite (reference applies the iterate)
    out, _, _, st = eng.ab_step([0.0], [60.0], [0.0], [0.0], [1e4], [10.0], [0.0])
    assert st.cpu().numpy()[0] == 1 and np.isfinite(out.cpu().numpy()).all()
    # zero-length calls are no-ops
    import torch
    e = torch.empty(0, dtype=torch.float64, device="cuda")
    eng.run_abmpc(e, e, e, torch.empty((3, 0), dtype=torch.float64, device="cuda"),
                  torch.empty((3, 0), dtype=torch.float64, device="cuda"))


def _route_case(tree, N, **kw):
    """A route with every feature of EstimateRouteAndComfortBounds switched on."""
    from eepacc_mpc_casadi_matlab_amd.settings import Settings, SetVehicleParameters, default_opt
    OPT = default_opt()
    OPT.update(slopes=np.array([[3.0, 40, 160], [-2.0, 300, 420]]),
               speedLimZones=np.array([[50.0, 0.0], [30.0, 150.0], [70.0, 400.0]]),
               curves=np.array([[-1 / 25.0, 90, 120], [1 / 60.0, 250, 300]]),
               stopLoc=np.array([200.0, 520.0]),
               TLLoc=np.array([[330.0, 5.0, 20.0, 25.0], [600.0, 0.0, 15.0, 15.0]]))
    OPT = Settings(OPT, tree=tree, N_hor=N)
    OPT.update(kw)
    return OPT, SetVehicleParameters(tree)


@pytest.mark.parametrize("tree", ["ABO", "ORIG"])
def test_route_features_and_estimator_modes(tree, torch_mod, lead_trace):
    """Speed-limit steps, curves, stops, traffic lights, road slope (non-constant: the sin/cos path of
    the plant), and the estimator modes 0 (constant velocity) and 2 (shifted previous solution,
    EstimateVehicleTrajectory.m:81-88) -- none of these has a reference golden, the oracle is the
    checker (closed loop, 90 steps)."""
    from oracle import Oracle
    for est in (dict(), dict(paramEstSetting=0, TVestSetting=0), dict(paramEstSetting=2)):
        OPT, V = _route_case(tree, 20, **est)
        B, n_steps = 4, 90
        sc = make_s2(B, n_steps, lead_trace["V_TO_2Hz"], seed=7)
        sc["s_tv"] = sc["s_tv"] + np.array([5.0, 60.0, 150.0, 1e4])[None, :]     # from tight following to a free road
        eng = _engine(OPT, V, 8)
        traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
        tr = traj.cpu().numpy(); st = status.cpu().numpy()
        orc = Oracle(OPT, V)
        tol = dict(s=1e-7, v=1e-8, a=1e-8, xi_v=1e-8, xi_h=1e-8, xi_s=1e-8, xi_f=1e-8, Fm=1e-4, Fb=1e-4)
        for i in range(B):
            ref, rst, _ = orc.run("ab", n_steps, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
            assert np.array_equal(rst != 0, st[:, i] != 0), (tree, est, i)
            ok = rst == 0
            for n, t in tol.items():
                assert np.abs(tr[ok, OUT[n], i] - ref[ok, OUT[n]]).max() < t, (tree, est, i, n)
        if est.get("paramEstSetting") == 2:       # the previous solution also travels between launches
            t1, _ = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][:37], sc["v_tv"][:37])
            t2, _ = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"][37:], sc["v_tv"][37:], resume=True)
            np.testing.assert_array_equal(np.concatenate([t1.cpu().numpy(), t2.cpu().numpy()]), tr)


def test_variable_time_steps(torch_mod, lead_trace):
    """Geometric Tvec (the commented option of ABO/Settings.m:120-121): general-T condensing scans."""
    from oracle import Oracle
    OPT, V, _, _ = make_case("ABO", 24)
    Ts, TsMax, N = 0.5, 1.5, 24
    OPT["Tvec"] = Ts * (TsMax / Ts) ** (np.arange(N) / (N - 1))
    B, n_steps = 3, 60
    sc = make_s2(B, n_steps, lead_trace["V_TO_2Hz"], seed=3)
    eng = _engine(OPT, V, 4)
    traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    tr = traj.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    orc = Oracle(OPT, V)
    for i in range(B):
        ref, rst, _ = orc.run("ab", n_steps, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
        assert rst.sum() == 0
        for n in ("s", "v", "Fm", "a", "xi_v", "xi_h", "xi_s", "xi_f"):
            assert np.abs(tr[:, OUT[n], i] - ref[:, OUT[n]]).max() < 10 * TOL[n], (i, n)


@pytest.mark.parametrize("case", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_reference_use_cases(case, torch_mod):
    """The predefined use cases of GetUseCase.m (route tables of Settings.m's useCaseNum switch), no
    lead vehicle (s_tv = inf, Main.m:77-80) except the recorded leads of 8, 9 and the cut-in scenario 10: closed loop against the
    oracle over the use case's own simulated time.  ORIG tree: its ABMPC has the speed-limit / curve /
    stop / traffic-light rows (ABO's CreateQP_AB.m:324-346 has them commented out)."""
    from oracle import Oracle
    from eepacc_mpc_casadi_matlab_amd.settings import Settings, SetVehicleParameters, default_opt
    o = default_opt(); o["useCaseNum"] = case
    if case in (8, 9):      # recorded lead vehicle (fixture extracted by tools/make_golden.py)
        rec = load_golden("argonne_61505019_lead")
        o["argonne_lead"] = (rec["t"], rec["v_mph"])
    OPT = Settings(o, tree="ORIG", N_hor=20)
    V = SetVehicleParameters("ORIG")
    n_steps = int(round(OPT["t_sim"] / OPT["Tvec"][0])) + 1
    if case in (8, 9, 10):
        s_tv, v_tv = np.asarray(OPT["s_tv"], dtype=np.float64), np.asarray(OPT["v_tv"], dtype=np.float64)
    else:
        s_tv, v_tv = np.full(n_steps, np.inf), np.zeros(n_steps)
    eng = _engine(OPT, V, 4)
    B = 2
    traj, status = eng.run_abmpc(np.full(B, OPT["s_init"]), np.full(B, OPT["v_init"]), np.full(B, OPT["a_minus1"]),
                                 np.repeat(s_tv[:, None], B, 1), np.repeat(v_tv[:, None], B, 1))
    tr = traj.cpu().numpy(); st = status.cpu().numpy()
    eng.synchronize()
    # run-to-run and instance-to-instance bit reproducibility first (a tolerance failure below is then a
    # deterministic distance to the oracle, not a race): same engine, second launch
    traj2, status2 = eng.run_abmpc(np.full(B, OPT["s_init"]), np.full(B, OPT["v_init"]), np.full(B, OPT["a_minus1"]),
                                   np.repeat(s_tv[:, None], B, 1), np.repeat(v_tv[:, None], B, 1))
    assert np.array_equal(traj2.cpu().numpy(), tr) and np.array_equal(status2.cpu().numpy(), st), case
    assert np.abs(tr[:, :, 0] - tr[:, :, 1]).max() == 0.0
    ref, rst, _ = Oracle(OPT, V).run("ab", n_steps, OPT["s_init"], OPT["v_init"], OPT["a_minus1"], s_tv.copy(), v_tv.copy())
    assert np.array_equal(rst != 0, st[:, 0] != 0)
    assert rst.sum() == 0
    assert np.isfinite(tr).all()
    # ORIG weights span 1e2 .. 1e7 (w_f): the two solvers stop at KKT points 1e-8 apart and the closed loop
    # amplifies that while crawling towards a stop line.  Largest distances measured on MI355X over the use
    # cases (tools/gpu_uc_margins.py -> profiles/r02_uc_margins.txt, final round-2 kernel): case 5 s 6.5e-8, v 3.3e-8,
    # a 6.5e-8, Fm 9.9e-5; case 12 v 2.0e-8, a 5.3e-8.  The tolerances keep a factor >= 4.5 over those (the round-1 values sat on a
    # 1.06e-7 reading of case 12 that no release build reproduces; tests/test_abi.py now pins the build flags).
    tol = dict(s=1e-6, v=3e-7, a=3e-7, xi_v=3e-7, xi_h=3e-7, xi_s=3e-7, xi_f=3e-7, Fm=1e-3, Fb=1e-3)
    # the long route (11) is compared in closed loop up to the approach of the stop line at 4000 m,
    # where a 1e-8 difference between the solvers is amplified by the loop (2 m after 800 more steps);
    # beyond it every step is checked as an open-loop QP at the oracle's states instead
    n_cl = 560 if case == 11 else n_steps
    for n, t in tol.items():
        assert np.abs(tr[:n_cl, OUT[n], 0] - ref[:n_cl, OUT[n]]).max() < t, (case, n)
    if case == 11:
        Ts = OPT["Tvec"][0]
        v = ref[:, OUT["v"]]
        a_prev = np.concatenate([[OPT["a_minus1"]], np.diff(v) / Ts])
        eng2 = _engine(OPT, V, n_steps)
        out, _, _, st2 = eng2.ab_step(ref[:, OUT["s"]].copy(), v.copy(), a_prev, Ts * np.arange(n_steps), s_tv.copy(), v_tv.copy(),
                                      np.zeros(n_steps), want_pred=False)
        o2 = out.cpu().numpy()
        assert int(st2.cpu().numpy().sum()) == 0
        for n, t in dict(xi_v=1e-6, xi_h=1e-6, xi_s=1e-6, xi_f=1e-6, a=1e-6, Fm=1e-2, Fb=1e-2).items():
            assert np.abs(o2[OUT[n]] - ref[:, OUT[n]]).max() < t, (case, "open loop", n)


@pytest.mark.parametrize("tree", ["ABO", "ORIG"])
def test_move_blocking(tree, torch_mod, lead_trace):
    """Mb != 0 (ABO/Settings.m:243-250, rows CreateQP_AB.m:282-288: a_k = a_{k-1} on blocked stages),
    done in the kernel as a change of variables; closed loop and per-step operator against the oracle,
    which keeps the equality rows."""
    from oracle import Oracle
    OPT, V, _, _ = make_case(tree, 20)
    OPT["Mb"] = np.array([0, 0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1, 0, 1, 1, 0, 1], dtype=np.int32)
    B, n_steps = 3, 80
    sc = make_s2(B, n_steps, lead_trace["V_TO_2Hz"], seed=2)
    eng = _engine(OPT, V, 4)
    traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    tr = traj.cpu().numpy(); st = status.cpu().numpy()
    orc = Oracle(OPT, V)
    tol = dict(s=1e-7, v=1e-8, a=1e-8, xi_v=1e-8, xi_h=1e-8, xi_s=1e-8, xi_f=1e-8, Fm=1e-4, Fb=1e-4)
    for i in range(B):
        ref, rst, _ = orc.run("ab", n_steps, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
        assert rst.sum() == 0 and st[:, i].sum() == 0
        for n, t in tol.items():
            assert np.abs(tr[:, OUT[n], i] - ref[:, OUT[n]]).max() < t, (tree, i, n)
    # predicted accelerations really are blocked: v_pred has equal increments inside a block
    a_prev = float((tr[40, OUT["v"], 0] - tr[39, OUT["v"], 0]) / 0.5)
    a_tv = float((sc["v_tv"][40, 0] - sc["v_tv"][39, 0]) / 0.5)
    inp = (float(tr[40, OUT["s"], 0]), float(tr[40, OUT["v"], 0]), a_prev, 20.0, float(sc["s_tv"][40, 0]), float(sc["v_tv"][40, 0]), a_tv)
    r = orc.ab_step(*inp)
    out, sp, vp, st1 = eng.ab_step(*[[x] for x in inp])
    vp = vp.cpu().numpy()[:, 0]
    assert np.abs(vp - r["v_pred"]).max() < 1e-7
    acc = np.diff(vp) / 0.5
    blocked = np.nonzero(OPT["Mb"])[0]
    assert np.abs(acc[blocked] - acc[blocked - 1]).max() < 1e-9


@pytest.mark.parametrize("N", [2, 3, 32, 33, 63])
def test_horizon_limits(N, torch_mod, lead_trace):
    """Shortest horizons, both sides of the small / large kernel configuration (32 | 33) and the maximum."""
    from oracle import Oracle
    OPT, V, _, _ = make_case("ABO", N)
    B, n_steps = 2, 12
    sc = make_s2(B, n_steps, lead_trace["V_TO_2Hz"], seed=4)
    eng = _engine(OPT, V, 2)
    traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    tr = traj.cpu().numpy()
    assert int(status.cpu().numpy().sum()) == 0
    orc = Oracle(OPT, V)
    for i in range(B):
        ref, rst, _ = orc.run("ab", n_steps, 0.0, float(sc["v0"][i]), 0.0, sc["s_tv"][:, i].copy(), sc["v_tv"][:, i].copy())
        assert rst.sum() == 0
        for n in ("s", "v", "Fm", "a", "xi_v", "xi_h", "xi_s", "xi_f"):
            assert np.abs(tr[:, OUT[n], i] - ref[:, OUT[n]]).max() < 10 * TOL[n], (N, i, n)


def test_host_wrapper_equals_device_path(torch_mod, lead_trace):
    """eepacc_run_abmpc_host (host numpy buffers in and out: the entry the MEX gateway mex/RunOpt_ABMPC.c
    calls) against the device-pointer path and the ABO golden."""
    OPT, V, s_tv, v_tv = make_case("ABO", 20)
    G = load_golden("abo_abmpc")
    eng = _engine(OPT, V, 4)
    B, n = 3, 200
    stv = np.repeat(s_tv[:n, None], B, 1); vtv = np.repeat(v_tv[:n, None], B, 1)
    th, sh = eng.run_abmpc_host(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
    td, sd = eng.run_abmpc(np.zeros(B), np.zeros(B), np.zeros(B), stv, vtv)
    assert np.array_equal(th, td.cpu().numpy()) and np.array_equal(sh, sd.cpu().numpy())
    assert sh.sum() == 0
    for nm in ("s", "v", "Fm", "xi_v", "xi_h"):
        assert np.abs(th[:, OUT[nm], 0] - G[nm + "_opt"][:n]).max() < TOL[nm], nm


def test_handoff_timeout_is_reported(torch_mod, lead_trace, monkeypatch):
    """Debug hook EEPACC_DEBUG_SPIN_LIMIT=0: a work unit whose predecessor is not yet published gives up at
    once.  The unit must not continue from stale state: its steps carry status 3, the rest of the launch is
    abandoned and eepacc_synchronize reports EEPACC_EDEVICE; after a reset the handle works again."""
    from eepacc_mpc_casadi_matlab_amd.engine import EepaccError
    OPT, V, _, _ = make_case("ABO", 20)
    B, n_steps = 64, 80
    sc = make_s2(B, n_steps, lead_trace["V_TO_2Hz"], seed=5)
    eng = _engine(OPT, V, B)
    good, gst = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    eng.synchronize()
    good = good.cpu().numpy()
    monkeypatch.setenv("EEPACC_DEBUG_SPIN_LIMIT", "0")
    traj, status = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    st = status.cpu().numpy()
    if (st == 3).any():
        with pytest.raises(EepaccError):
            eng.synchronize()
        ok = st != 3
        # steps that were computed before the failure are the regular results
        first_bad = np.argmax(st == 3, axis=0)
        for i in range(B):
            kb = first_bad[i] if (st[:, i] == 3).any() else n_steps
            assert np.array_equal(traj.cpu().numpy()[:kb, :, i], good[:kb, :, i])
            assert (st[kb:, i] == 3).all()
    else:       # every predecessor happened to be published in time
        eng.synchronize()
    monkeypatch.delenv("EEPACC_DEBUG_SPIN_LIMIT")
    again, ast = eng.run_abmpc(sc["s0"], sc["v0"], sc["a_minus1"], sc["s_tv"], sc["v_tv"])
    eng.synchronize()
    assert np.array_equal(again.cpu().numpy(), good) and int(ast.cpu().numpy().sum()) == 0
