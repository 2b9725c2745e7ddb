"""Key-figure report (ABO/Main.m:131-263) on the reference's saved ABMPC / FBMPC solutions."""
import numpy as np

from conftest import load_golden, make_case
from eepacc_mpc_casadi_matlab_amd.report import kpi_report, format_report, InterpPWA


def test_kpis_of_the_saved_solutions():
    OPT, V, _, _ = make_case("ABO", 20)
    G = dict(load_golden("abo_abmpc"))
    k = kpi_report(G, OPT)
    assert abs(k["distance_km"] - 3.091279) < 1e-5                 # 3091.279 m travelled
    assert abs(k["energy_kWh"] - float(G["E_opt"][-1]) / 3.6e6) < 1e-12
    assert k["bad_exit_messages"] == 0
    # the default route is 3.5 km long at most: the run ends before the cut-off distance (Main.m:158-160)
    assert k["cutoff_index"] == 870 and k["travel_time_at_cutoff_s"] == 435.0
    assert k["a_max"] == G["a_opt"][:870].max() and k["j_min"] == G["j_opt"][:870].min()
    assert abs(k["a_rms"] - np.sqrt(np.mean(G["a_opt"][:870] ** 2))) < 1e-15
    # cut-off inside the run: index of the bracketing pair, values one sample before it (Main.m:232,245)
    OPT2 = dict(OPT); OPT2["cutOffDist"] = 1500.0
    k2 = kpi_report(G, OPT2)
    i = int(k2["cutoff_index"])
    assert G["s_opt"][i - 1] < 1500.0 < G["s_opt"][i]
    assert k2["energy_at_cutoff_kWh"] == G["E_opt"][i - 2] / 3.6e6
    assert abs(k2["speed_limit_error_at_cutoff"] - (50 / 3.6 - G["v_opt"][i - 2])) < 1e-12      # 50 km/h zone from 1000 m
    txt = format_report("Acceleration-based MPC", k2, OPT2)
    assert "Travel time at 1.5 km" in txt and "bad exit messages" in txt
    F = dict(load_golden("abo_fbmpc"))
    assert kpi_report(F, OPT)["distance_km"] > 3.0


def test_interp_pwa():
    assert InterpPWA(-5, [0, 10], [1, 3]) == 1 and InterpPWA(50, [0, 10], [1, 3]) == 3
    assert abs(InterpPWA(2.5, [0, 10], [1, 3]) - 1.5) < 1e-15


def test_fuel_economy_of_saved_solutions():
    """ABO/Custom_plots.m:73-155: L/100 km of the runs with and without the fuel term and of the lead trace itself.  The
    reference prints these numbers without storing them, so the check is the formula on hand-computable inputs plus
    the ordering the script was written to show (the fuel-optimised run uses less than the run without the term)."""
    import numpy as np
    from conftest import load_golden
    from eepacc_mpc_casadi_matlab_amd.report import fuel_economy, fuel_economy_of_speed_trace
    from eepacc_mpc_casadi_matlab_amd.settings import SetVehicleParameters
    V = SetVehicleParameters("ABO")
    # constant 10 m/s for 100 s: TW = (F0 + F2 v^2) R_w, FC = p00 + p10 v + p01 TW (above the 0.25 g/s floor)
    n = 201
    sol = dict(a_opt=np.zeros(n), v_opt=np.full(n, 10.0), s_opt=10.0 * 0.5 * np.arange(n))
    fe = fuel_economy(sol, V)
    TW = (V["F0"] + V["F2"] * 100.0) * V["R_w"]
    FC = max(0.25, V["p00"] + V["p10"] * 10.0 + V["p01"] * TW)
    assert fe["FC"][1] == FC and fe["TW_opt"][1] == TW and fe["FC"][0] == 0.0
    assert abs(fe["FE_L_per_100km"] - FC / 1000 * 0.5 * (n - 1) / 0.835 / (sol["s_opt"][-1] / 1000) * 100) < 1e-12
    fc = fuel_economy(load_golden("abo_abmpc_fcopt"), V)["FE_L_per_100km"]
    nofc = fuel_economy(load_golden("abo_abmpc_nofcopt"), V)["FE_L_per_100km"]
    lead = fuel_economy_of_speed_trace(load_golden("lead_TO01_EAD")["V_TO_2Hz"], V)
    # 9.75 (fuel term), 10.20 (no fuel term), 10.58 L/100 km (lead trace): the saving the reference's script shows
    assert 9.7 < fc < 9.8 and 10.1 < nofc < 10.3 and 10.5 < lead < 10.7
