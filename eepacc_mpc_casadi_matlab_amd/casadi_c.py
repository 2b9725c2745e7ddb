"""Settings file of the B4 adaptor (include/eepacc_casadi_c.h): what `casadi_c_push_file` of libeepacc reads in
place of a serialized CasADi Function (ABO/casadi_fun.c:61)."""
from __future__ import annotations

from typing import Any, Dict

import numpy as np

from ._abi import _VEH_FIELDS, _VEH_OPTIONAL


def write_config(path: str, OPT: Dict[str, Any], V: Dict[str, float]) -> None:
    """Flat "key values..." text of the OPTsettings / vehicle fields of include/eepacc.h."""
    def arr(x):
        return " ".join(repr(float(v)) for v in np.asarray(x, dtype=np.float64).ravel())
    W_AB = np.asarray(OPT["W_AB"], dtype=np.float64).ravel()
    fuel = (2 if OPT.get("fuel_map", "EFF") == "ICE" else 1) if W_AB.size == 7 else 0
    if W_AB.size == 6:
        W_AB = np.concatenate([[0.0], W_AB])
    lines = ["# eepacc casadi_c settings file", "N_hor %d" % int(OPT["N_hor"]), "Tvec " + arr(OPT["Tvec"]),
             "Mb " + arr(OPT.get("Mb", np.zeros(int(OPT["N_hor"])))), "W_AB " + arr(W_AB), "W_FB " + arr(OPT["W_FB"]),
             "ab_fuel_term %d" % fuel,
             "ab_route_rows %d" % int(OPT.get("ab_route_rows", 1 if OPT.get("tree", "ABO") == "ORIG" else 0))]
    for k in ("tau_min", "h_min", "s_goal", "paramEstSetting", "TVestSetting", "tConstACC_ego", "tConstACC_tar",
              "N_integratePlant", "solverToUse", "stopRefDist", "stopRefVelSlope", "stopVel", "TLstopVel",
              "TLStopRegionSize", "alpha_TTL"):
        lines.append("%s %r" % (k, float(OPT[k])))
    lines.append("FBuseTaylor %d" % int(bool(OPT["FBuseTaylor"])))
    for k in ("b_quadr", "b_fifthOrder", "s_speedLim", "v_speedLim", "s_curv", "curvature", "s_slope", "slope"):
        lines.append(k + " " + arr(OPT[k]))
    if np.asarray(OPT.get("stopLoc", [])).size:
        lines.append("stopLoc " + arr(OPT["stopLoc"]))
    if np.asarray(OPT.get("TLLoc", [])).size:
        lines.append("TLLoc " + arr(np.asarray(OPT["TLLoc"], dtype=np.float64).reshape(-1, 4)))
    for f in _VEH_FIELDS:
        if f in V or f not in _VEH_OPTIONAL:
            lines.append("vehicle.%s %r" % (f, float(V[f])))
    if "upSpd" in V and "tau_gb" in V:
        lines.append("vehicle.upSpd " + arr(V["upSpd"]))
        lines.append("vehicle.tau_gb " + arr(V["tau_gb"]))
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
