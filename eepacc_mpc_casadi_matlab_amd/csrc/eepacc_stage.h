// eepacc_stage.h -- per-stage device functions shared by the ABMPC and FBMPC kernels:
// trajectory estimator (A2), route/comfort bounds (A3), PWA interpolation and the RK4 plant (A9).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "eepacc_device.h"

namespace eepacc {

// ABO/Functions/MPCs/EstimateVehicleTrajectory.m:55-80 (modes 0 and 1), value for stage `lane`
__device__ inline void estimate_traj(const DevCfg& C, int mode, double tConstACC, double s0, double v0, double a0,
                              int lane, double& s_est, double& v_est) {
    const int N = C.N;
    const int i = lane <= N ? lane : N;
    if (mode == 0) {
        s_est = s0 + C.tau[i] * v0;      // sum_{j<i} Tvec[j]*v0 (one rounding vs repeated adds)
        v_est = v0;
        return;
    }
    // mode 1: constant acceleration for the first steps, sequentially as the reference does
    double s = s0, v = v0, ms = s0, mv = v0;
    int last = 0;
    for (int j = 1; j <= N; ++j) {
        const double Ts = C.Tvec[j - 1];
        const bool idx_ok = (double)(j + 1) <= tConstACC / Ts;
        if (!idx_ok && C.const_T) break;
        const double vprev = v;
        if (idx_ok && v + Ts * a0 > 0.0) v = v + Ts * a0;
        s = s + Ts * vprev;
        last = j;
        if (j == i) { ms = s; mv = v; }
    }
    if (i > last) {
        // remaining stages: constant velocity v from stage `last` on
        mv = v;
        ms = s + v * (C.tau[i] - C.tau[last]);
    }
    s_est = ms; v_est = mv;
}

// ABO/Functions/MPCs/EstimateRouteAndComfortBounds.m:89-171 for one stage
__device__ inline void route_bounds(const DevCfg& C, double s_est, double v_est, double t0, int i /*0-based stage*/,
                             double& v_lim, double& v_curv, double& v_stop, double& v_TL,
                             double& a_min, double& a_max, double& j_min, double& j_max) {
    v_lim = 0.0;
    for (int j = 0; j < C.n_speedLim; ++j) {
        if (j == C.n_speedLim - 1) v_lim = C.s_speedLim[C.n_speedLim - 1];     // sic (:93)
        else if (s_est >= C.s_speedLim[j] && s_est < C.s_speedLim[j + 1]) { v_lim = C.v_speedLim[j]; break; }
    }
    v_curv = 0.0;
    for (int j = 0; j < C.n_curv; ++j) {
        if (j == C.n_curv - 1) v_curv = C.vcurv_tab[C.n_curv - 1];
        else if (s_est > C.s_curv[j] && s_est < C.s_curv[j + 1]) { v_curv = C.vcurv_tab[j]; break; }
    }
    v_stop = 1e5;
    for (int j = 0; j < C.n_stop; ++j) {
        double dist = fabs(C.stopLoc[j] - s_est);
        if (dist < C.stopRefDist) v_stop = dist * C.stopRefVelSlope + C.stopVel;
    }
    v_TL = 1e5;
    for (int j = 0; j < C.n_TL; ++j) {
        const double* TL = &C.TLLoc[4 * j];
        double x = t0 + (double)(i + 1) * C.Tvec[i] - TL[1];
        double mm = TL[2] + TL[3];
        double md = (mm == 0.0) ? x : x - floor(x / mm) * mm;
        if (md < TL[2]) {
            double d = TL[0] - s_est;
            if (fabs(d) < C.stopRefDist) {
                if (d < 0.0) v_TL = fabs(d) * C.stopRefVelSlope + C.TLstopVel;
                else if (fabs(d) < C.TLStopRegionSize) v_TL = C.TLstopVel;
                else v_TL = fabs(d - C.stopVel) * C.stopRefVelSlope + C.TLstopVel;
            }
        }
    }
    if (C.bl_mode) {
        // baseline limits, EstimateRouteAndComfortBounds.m:173-189 (MPCtype 1)
        double a, j;
        if (v_est < 5.0) { a = C.bl_aLo; j = C.bl_jLo; }
        else if (v_est < 20.0) {
            a = (4.0 * C.bl_aLo - C.bl_aHi) / 3.0 + (C.bl_aHi - C.bl_aLo) / 15.0 * v_est;
            j = (4.0 * C.bl_jLo - C.bl_jHi) / 3.0 + (C.bl_jHi - C.bl_jLo) / 15.0 * v_est;
        } else { a = C.bl_aHi; j = C.bl_jHi; }
        a_min = -a; a_max = a; j_min = -j; j_max = j;
        return;
    }
    if (v_est < 5.0) { a_min = -5.0; a_max = 4.0; j_min = -5.0; j_max = 5.0; }
    else if (v_est < 20.0) {
        a_min = -5.5 + v_est / 10.0; a_max = 14.0 / 3.0 - 2.0 * v_est / 15.0;
        j_min = -35.0 / 6.0 + v_est / 6.0; j_max = 35.0 / 6.0 - v_est / 6.0;
    } else { a_min = -3.5; a_max = 2.0; j_min = -2.5; j_max = 2.5; }
}

// ABO/Functions/PWA_function_manipulation/InterpPWA.m:14-27
__device__ inline double interp_pwa(double d, const double* doms, const double* vals, int n) {
    if (d < doms[0]) return vals[0];
    if (d > doms[n - 1]) return vals[n - 1];
    for (int i = 0; i < n - 1; ++i)
        if (d >= doms[i] && d <= doms[i + 1]) {
            double f = (d - doms[i]) / (doms[i + 1] - doms[i]);
            return vals[i] + f * (vals[i + 1] - vals[i]);
        }
    return vals[n - 1];
}

__device__ __forceinline__ void slope_trig(const DevCfg& C, double s, double& sn, double& cs) {
    if (C.const_slope) { sn = C.sin_theta0; cs = C.cos_theta0; return; }
    double th = interp_pwa(s, C.s_slope, C.slope, C.n_slope);
    sn = sin(th); cs = cos(th);
}

// ABO/Functions/MPCs/RunPlantModel.m:27-44
__device__ inline void plant_rk4(const DevCfg& C, double s, double v, double u, double& s1, double& v1) {
    const int Mi = C.N_integratePlant;
    const double DT = C.Tvec[0] / Mi;
    const double ilm = 1.0 / C.lambda / C.m;
    double x0 = s, x1 = v;
    for (int k = 0; k < Mi; ++k) {
        double sn, cs;
        slope_trig(C, x0, sn, cs);
        double k10 = x1, k11 = ilm * (u - C.zeta_a * x1 * x1 - C.c_r * C.m * C.g * cs - C.m * C.g * sn);
        double y0 = x0 + DT / 2 * k10, y1 = x1 + DT / 2 * k11;
        slope_trig(C, y0, sn, cs);
        double k20 = y1, k21 = ilm * (u - C.zeta_a * y1 * y1 - C.c_r * C.m * C.g * cs - C.m * C.g * sn);
        y0 = x0 + DT / 2 * k20; y1 = x1 + DT / 2 * k21;
        slope_trig(C, y0, sn, cs);
        double k30 = y1, k31 = ilm * (u - C.zeta_a * y1 * y1 - C.c_r * C.m * C.g * cs - C.m * C.g * sn);
        y0 = x0 + DT * k30; y1 = x1 + DT * k31;
        slope_trig(C, y0, sn, cs);
        double k40 = y1, k41 = ilm * (u - C.zeta_a * y1 * y1 - C.c_r * C.m * C.g * cs - C.m * C.g * sn);
        x0 = x0 + DT / 6 * (k10 + 2 * k20 + 2 * k30 + k40);
        x1 = x1 + DT / 6 * (k11 + 2 * k21 + 2 * k31 + k41);
    }
    s1 = x0; v1 = x1;
}

}  // namespace eepacc
