// eepacc_device.h -- device-side configuration shared by the HIP kernels and the host C-ABI.
#pragma once
#include <stdint.h>

namespace eepacc {

constexpr int kWave = 64;
constexpr int kMaxKnots = 64;     // route table knots kept in the device config (use case 11 needs 42)
constexpr int kMaxStops = 16;
constexpr int kMaxTL = 8;
constexpr int kMaxN = 63;         // stage k lives on lane k, terminal stage on lane N

// Row types of one stage of the ABMPC QP.  Order/meaning follow the reference row list
// ABO/Functions/MPCs/CreateQP_AB.m:256-371 (+ the four ORIG rows :307-329); every row is
//    al*s_k + be*v_k + ga*a_k + de*a_{k-1} - xi_group <= b
enum RowType : int {
    R_SLO = 0, R_SHI, R_VLO, R_VHI,                 // hard state bounds  (:256-263)
    R_AMAX, R_AMIN, R_JMAX, R_JMIN, R_VLIM, R_VCURV, // group F (xi_f)     (:292-322, ORIG :307-317)
    R_SAFE1, R_SAFE2, R_VSTOP, R_VTL,                // group S (xi_s)     (:355-362, ORIG :319-329)
    R_VINC,                                          // group V (xi_v)     (:349-352)
    R_HWP,                                           // group H (xi_h)     (:365-371), quadratic slack
    kNumRowTypes = 16
};
enum Group : int { G_NONE = 0, G_F = 1, G_S = 2, G_V = 3, G_H = 4 };

struct DevCfg {
    int32_t N;
    int32_t ab_fuel_term, ab_route_rows;
    int32_t paramEstSetting, TVestSetting;
    int32_t N_integratePlant;
    int32_t const_T;              // 1: all Tvec equal
    int32_t const_slope;          // 1: slope table is constant -> theta fixed
    int32_t mb_any;               // move blocking requested (ABO/Settings.m:243-250): a_k = a_{k-1} where Mb[k] = 1
    int32_t max_iter;
    double Tvec[kMaxN + 1];
    double tau[kMaxN + 2];        // tau[k] = sum_{i<k} Tvec[i]
    double w_FC, w_a, w_j, w_v, w_h, w_s, w_f;
    double tau_min, h_min, s_goal;
    double tConstACC_ego, tConstACC_tar;
    // vehicle
    double m, lambda, g, zeta_a, c_r, mu, L, L_f, h_g, phi, T_m_max, P_m_max, eta_TF, omega_m_r, v_max;
    double p01, p10, F2;
    double sin_theta0, cos_theta0, theta0;
    // objective constants (ABO/.../CreateQP_AB.m:162-187)
    double cq;                    // w_FC*p01*F2
    double glin_v;                // w_FC*p10
    double glin_a;                // w_FC*p01*lambda*m
    // route tables (GenerateUseCase.m) -- ascending knots
    int32_t n_speedLim, n_curv, n_slope, n_stop, n_TL;
    double s_speedLim[kMaxKnots], v_speedLim[kMaxKnots];
    double s_curv[kMaxKnots], vcurv_tab[kMaxKnots];      // alpha_TTL*|curvature|^(-1/3)
    double s_slope[kMaxKnots], slope[kMaxKnots];
    double stopLoc[kMaxStops];
    double TLLoc[kMaxTL * 4];
    double stopRefDist, stopRefVelSlope, stopVel, TLstopVel, TLStopRegionSize;
    double b5[21];                // fifth-order power surface
    // FBMPC (ABO/Settings.m:31-46, 229): weights [w_P,w_a,w_j,w_v,w_h,w_s,w_f], quadratic power fit
    double fb_w[7];
    double b_quadr[6];
    int32_t FBuseTaylor, fb_pad;
    const double* Hinv;           // device, [N][N] row-major, inverse of the a-space Hessian
    double* pred;                 // device, [max_batch][2][64]: previous predicted s, v (paramEstSetting 2)
    // move blocking: stage k takes the acceleration of its block leader mb_lead[k]; a leader's block ends at mb_end[k]
    int32_t mb_lead[kMaxN + 1], mb_end[kMaxN + 1], mb_maxlen, mb_pad;
    // FBMPC row layout: stage k owns rows fb_row0[k] .. fb_row0[k+1]-1 (26, or 28 with its two blocked-move rows)
    int32_t fb_row0[kMaxN + 2];
    int32_t mb_mask[kMaxN + 1];
    // baseline controller (RunOpt_BLMPC / CreateQP_BL): one slack group, travel incentive -w_v sum v_k
    int32_t bl_mode, bl_prox_max;  // bl_prox_max: proximal re-centrings of the LP after its first solve (0: none)
    double bl_eps;                // proximal curvature of an LP (w_a = w_j = 0); excluded from the reported cost
    double state_tol;             // tolerance of the hard bounds on the measured state (constant rows of stage 0)
    double bl_aLo, bl_aHi, bl_jLo, bl_jHi;   // BL_a_LimLowVel, BL_a_LimHighVel, BL_j_LimLowVel, BL_j_LimHighVel
    // ICE-map fuel term (ab_fuel_term = 2, CreateQP_AB.m:154-159): per stage cq_k = ice_cq / tau_k, lv_k = ice_lv tau_k,
    // la_k = ice_la / tau_k with the gear ratio tau_k = ice_gb[first g: v_est < ice_up[g]] (LUTgearshift.m:17-41)
    double ice_cq, ice_lv, ice_la, ice_up[7], ice_gb[8];
    double* hb;                   // device, per-wave NS x NS scratch: the step's base inverse of the ICE variant
};

}  // namespace eepacc
