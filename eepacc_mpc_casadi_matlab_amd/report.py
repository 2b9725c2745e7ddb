"""Key figures of a closed-loop run: the numbers ABO/Main.m:131-263 prints for one controller.

``kpi_report(optSol, OPTsettings)`` takes the struct returned by ``RunOpt_ABMPC`` / ``RunOpt_FBMPC``
(or a saved solution with the same fields) and returns them as a dict; ``format_report`` renders the
text block of Main.m for one controller.  Host-side numpy only.
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np


def InterpPWA(d: float, doms, vals) -> float:
    """ABO/Functions/PWA_function_manipulation/InterpPWA.m:14-27."""
    doms = np.asarray(doms, dtype=np.float64); vals = np.asarray(vals, dtype=np.float64)
    if d < doms[0]:
        return float(vals[0])
    if d > doms[-1]:
        return float(vals[-1])
    for i in range(doms.size - 1):
        if doms[i] <= d <= doms[i + 1]:
            return float(vals[i] + (d - doms[i]) / (doms[i + 1] - doms[i]) * (vals[i + 1] - vals[i]))
    return float(vals[-1])


def _rms(x) -> float:
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.mean(x * x))) if x.size else 0.0


def kpi_report(sol: Dict[str, Any], OPT: Dict[str, Any]) -> Dict[str, float]:
    s = np.asarray(sol["s_opt"], dtype=np.float64).ravel()
    v = np.asarray(sol["v_opt"], dtype=np.float64).ravel()
    a = np.asarray(sol["a_opt"], dtype=np.float64).ravel()
    j = np.asarray(sol["j_opt"], dtype=np.float64).ravel()
    E = np.asarray(sol["E_opt"], dtype=np.float64).ravel()
    Ts = float(np.asarray(OPT["Tvec"]).ravel()[0])
    cut = float(OPT["cutOffDist"])
    # first sample pair that brackets the cut-off distance (1-based index as in Main.m:150-161)
    ind = 0
    for i in range(1, s.size):
        if s[i - 1] < cut and s[i] > cut:
            ind = i
            break
    if ind == 0:
        ind = s.size - 1
    vlim = InterpPWA(cut, OPT["s_speedLim"], OPT["v_speedLim"])                      # :133
    k = ind - 1                                                                     # MATLAB (ind-1), 1-based -> 0-based ind-2
    return {
        "bad_exit_messages": float(np.sum(np.asarray(sol["exitMessage"]))),           # :210
        "distance_km": s[-1] / 1e3,                                                  # :220
        "energy_kWh": E[-1] / 3.6e6,                                                 # :226
        "cutoff_index": float(ind),
        "speed_limit_error_at_cutoff": vlim - v[k - 1],                              # :232
        "energy_at_cutoff_kWh": E[k - 1] / 3.6e6,                                    # :245
        "travel_time_at_cutoff_s": 0.1 * round(ind * Ts * 10),                       # :238
        "a_max": float(a[:ind].max()), "a_min": float(a[:ind].min()), "a_rms": _rms(a[:ind]),       # :257
        "j_max": float(j[:ind].max()), "j_min": float(j[:ind].min()), "j_rms": _rms(j[:ind]),       # :261
    }


def format_report(name: str, k: Dict[str, float], OPT: Dict[str, Any]) -> str:
    cut = float(OPT["cutOffDist"]) / 1e3
    return "\n".join([
        f"=== RESULTS of UC{OPT.get('useCaseNum', 0)} ===", "",
        "Feasibility:", f"   {name}: {k['bad_exit_messages']:g} bad exit messages", "",
        "Distance traveled:", f"   {name}: {k['distance_km']:.5g} km", "",
        "Energy consumption:", f"   {name}: {k['energy_kWh']:.5g} kWh", "",
        f"Error relative to the speed limit at {cut:g} km:", f"   {name}: {k['speed_limit_error_at_cutoff']:.5g} m/s", "",
        f"Energy consumption at {cut:g} km:", f"   {name}: {k['energy_at_cutoff_kWh']:.5g} kWh", "",
        f"Travel time at {cut:g} km:", f"   {name}: {k['travel_time_at_cutoff_s']:g} s", "",
        f"Comfort metrics at {cut:g} km:",
        f"   {name}: max. a = {k['a_max']:.5g}m/s2, min. a = {k['a_min']:.5g}m/s2, rms. a = {k['a_rms']:.5g}m/s2",
        f"   {name}: max. j = {k['j_max']:.5g}m/s3, min. j = {k['j_min']:.5g}m/s3, rms. j = {k['j_rms']:.5g}m/s3", ""])


def fuel_economy(sol: Dict[str, Any], V: Dict[str, float], Ts: float = 0.5) -> Dict[str, Any]:
    """Fuel consumption of a closed-loop run with the linear fuel map of the ABO tree (ABO/Custom_plots.m:73-107):
    wheel torque TW = max(0, (lambda m a + F0 + F2 v^2) R_w), fuel flow FC = max(0.25, p00 + p10 v + p01 TW) [g/s],
    cumulative fuel [kg] with the sample time Ts (0.5 s in the reference) and fuel economy [L/100 km] at a density of
    0.835 kg/L.  sol: struct with a_opt, v_opt, s_opt (RunOpt_ABMPC / a saved solution)."""
    a = np.asarray(sol["a_opt"], dtype=np.float64).ravel()
    v = np.asarray(sol["v_opt"], dtype=np.float64).ravel()
    s = np.asarray(sol["s_opt"], dtype=np.float64).ravel()
    TW = np.maximum(0.0, (V["lambda"] * V["m"] * a + V["F0"] + V["F2"] * v * v) * V["R_w"])
    FC = np.maximum(0.25, V["p00"] + V["p10"] * v + V["p01"] * TW)
    TW[0] = 0.0; FC[0] = 0.0                        # the loop starts at i = 2 (:81): the first sample stays zero
    FC_tot = np.cumsum(FC / 1000.0 * Ts)
    return {"TW_opt": TW, "FC": FC, "FC_tot_kg": FC_tot,
            "FE_L_per_100km": float(np.max(FC_tot / 0.835) / np.max(s / 1000.0) * 100.0)}


def fuel_economy_of_speed_trace(V_2Hz, V: Dict[str, float], Ts: float = 0.5) -> float:
    """The same figure for a vehicle that drives the lead trace itself (ABO/Custom_plots.m:113-155, `FE_lead`;
    the gear-dependent quantities of that block do not enter the fuel map)."""
    v = np.asarray(V_2Hz, dtype=np.float64).ravel().copy()
    v = np.concatenate([v, [0.0]]) if v.size == 870 else v              # V_TO(871,1) = 0 (:117)
    a = np.concatenate([[0.0], np.diff(v) / Ts])
    s = np.concatenate([[0.0], np.cumsum(v[1:] * Ts)])
    return fuel_economy({"a_opt": a, "v_opt": v, "s_opt": s}, V, Ts)["FE_L_per_100km"]
