/*
 * eepacc_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Line-for-line fp64 restatement of the reference's per-step MPC pipeline:
 *   measurement -> EstimateVehicleTrajectory -> CreateQP_{AB,FB} (sparse-form dense
 *   matrices) -> TransformToDenseFormulation (literal loops + literal dense products) ->
 *   dense QP (qp_dense.c) -> z = Psi x + d -> force allocation -> RK4 plant -> post.
 * Each function cites the reference lines it follows (ABO/ = ACCMPC-ABO_CasADi/).
 * Pinned against the reference's saved solutions: tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "oracle.h"

#define IDX(i, j, ld) ((size_t)(i) * (size_t)(ld) + (size_t)(j))
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static double matlab_mod(double a, double m) {
    if (m == 0.0) return a;
    return a - floor(a / m) * m;
}

/* ABO/Functions/PWA_function_manipulation/InterpPWA.m:14-27 */
double orc_interp_pwa(double d, const double* doms, const double* vals, int n) {
    if (d < doms[0]) return vals[0];
    if (d > doms[n - 1]) return vals[n - 1];
    for (int i = 0; i < n - 1; ++i)
        if (d >= doms[i] && d <= doms[i + 1]) {
            double f = (d - doms[i]) / (doms[i + 1] - doms[i]);
            return vals[i] + f * (vals[i + 1] - vals[i]);
        }
    return vals[n - 1];
}

/* ABO/Functions/Other/GetMotorPower_FifthOrderSurface.m:16-20 */
double orc_motor_power_fifth(double x, double y, const double* b) {
    double x2 = x * x, x3 = x2 * x, x4 = x3 * x, x5 = x4 * x;
    double y2 = y * y, y3 = y2 * y, y4 = y3 * y, y5 = y4 * y;
    return b[0] + b[1] * x + b[2] * y + b[3] * x2 + b[4] * x * y + b[5] * y2 + b[6] * x3 +
           b[7] * x2 * y + b[8] * x * y2 + b[9] * y3 + b[10] * x4 + b[11] * x3 * y +
           b[12] * x2 * y2 + b[13] * x * y3 + b[14] * y4 + b[15] * x5 + b[16] * x4 * y +
           b[17] * x3 * y2 + b[18] * x2 * y3 + b[19] * x * y4 + b[20] * y5;
}

/* ABO/Functions/MPCs/EstimateVehicleTrajectory.m:55-88 */
void orc_estimate_vehicle_trajectory(const eepacc_settings* S, int estSetting, double s_curr,
                                     double v_curr, double a_curr, const double* s_prev_sol,
                                     const double* v_prev_sol, double* s_est, double* v_est) {
    int N = S->N_hor;
    int mode = estSetting == 0 ? S->paramEstSetting : S->TVestSetting;      /* :32-38 */
    double tConstACC = estSetting == 1 ? S->tConstACC_tar : S->tConstACC_ego; /* :41-46 */
    for (int i = 0; i <= N; ++i) { s_est[i] = 0.0; v_est[i] = 0.0; }
    if (mode == 0) {                                                         /* :55-64 */
        s_est[0] = s_curr;
        for (int i = 1; i <= N; ++i) s_est[i] = s_est[i - 1] + S->Tvec[i - 1] * v_curr;
        for (int i = 0; i <= N; ++i) v_est[i] = v_curr;
    } else if (mode == 1) {                                                  /* :65-80 */
        s_est[0] = s_curr;
        v_est[0] = v_curr;
        for (int i = 1; i <= N; ++i) {
            double Ts = S->Tvec[i - 1];
            /* MATLAB index is i+1 */
            if ((double)(i + 1) <= tConstACC / Ts && v_est[i - 1] + Ts * a_curr > 0.0)
                v_est[i] = v_est[i - 1] + Ts * a_curr;
            else
                v_est[i] = v_est[i - 1];
            s_est[i] = s_est[i - 1] + Ts * v_est[i - 1];
        }
    } else {                                                                 /* :81-88 */
        double Ts = S->Tvec[N - 1];
        s_est[0] = s_curr;
        v_est[0] = v_curr;
        for (int i = 1; i < N; ++i) { s_est[i] = s_prev_sol[i + 1]; v_est[i] = v_prev_sol[i + 1]; }
        s_est[N] = s_prev_sol[N] + Ts * v_prev_sol[N];
        v_est[N] = v_prev_sol[N];
    }
}

/* ABO/Functions/MPCs/EstimateRouteAndComfortBounds.m:63-208, MPCtype == 0 */
void orc_estimate_route_and_comfort_bounds(const eepacc_settings* S, const double* s_est,
                                           const double* v_est, double t_0, double* slope_est,
                                           double* v_lim_max, double* v_stop_max, double* v_TL_max,
                                           double* v_curv_max, double* a_min_est, double* a_max_est,
                                           double* j_min_est, double* j_max_est) {
    int N = S->N_hor;
    for (int i = 0; i < N; ++i) {
        /* slope :72-86 -- the inner loop always breaks at j == 1 */
        slope_est[i] = (S->n_slope == 1) ? S->slope[S->n_slope - 1] : S->slope[0];
        /* speed limit :89-99 (falls through to s_speedLim(end) at the last knot, :93) */
        v_lim_max[i] = 0.0;
        for (int j = 0; j < S->n_speedLim; ++j) {
            if (j == S->n_speedLim - 1) {
                v_lim_max[i] = S->s_speedLim[S->n_speedLim - 1];
            } else if (s_est[i] >= S->s_speedLim[j] && s_est[i] < S->s_speedLim[j + 1]) {
                v_lim_max[i] = S->v_speedLim[j];
                break;
            }
        }
        /* curve :102-112 */
        v_curv_max[i] = 0.0;
        for (int j = 0; j < S->n_curv; ++j) {
            if (j == S->n_curv - 1) {
                v_curv_max[i] = S->alpha_TTL * pow(fabs(S->curvature[S->n_curv - 1]), -1.0 / 3.0);
            } else if (s_est[i] > S->s_curv[j] && s_est[i] < S->s_curv[j + 1]) {
                v_curv_max[i] = S->alpha_TTL * pow(fabs(S->curvature[j]), -1.0 / 3.0);
                break;
            }
        }
        /* stops :115-123 */
        v_stop_max[i] = 1e5;
        for (int j = 0; j < S->n_stop; ++j) {
            double dist = fabs(S->stopLoc[j] - s_est[i]);
            if (dist < S->stopRefDist) v_stop_max[i] = dist * S->stopRefVelSlope + S->stopVel;
        }
        /* traffic lights :126-143 */
        v_TL_max[i] = 1e5;
        for (int j = 0; j < S->n_TL; ++j) {
            const double* TL = &S->TLLoc[4 * j];
            if (matlab_mod(t_0 + (double)(i + 1) * S->Tvec[i] - TL[1], TL[2] + TL[3]) < TL[2]) {
                double distToTL = TL[0] - s_est[i];
                if (fabs(distToTL) < S->stopRefDist) {
                    if (distToTL < 0.0)
                        v_TL_max[i] = fabs(distToTL) * S->stopRefVelSlope + S->TLstopVel;
                    else if (fabs(distToTL) < S->TLStopRegionSize)
                        v_TL_max[i] = S->TLstopVel;
                    else
                        v_TL_max[i] = fabs(distToTL - S->stopVel) * S->stopRefVelSlope + S->TLstopVel;
                }
            }
        }
        if (S->bl_mode) {                       /* baseline limits :173-189 (MPCtype 1) */
            double aL = S->BL_a_LimLowVel, aH = S->BL_a_LimHighVel, jL = S->BL_j_LimLowVel, jH = S->BL_j_LimHighVel, a, j;
            if (v_est[i] < 5.0) { a = aL; j = jL; }
            else if (v_est[i] < 20.0) {
                a = (4.0 * aL - aH) / 3.0 + (aH - aL) / 15.0 * v_est[i];
                j = (4.0 * jL - jH) / 3.0 + (jH - jL) / 15.0 * v_est[i];
            } else { a = aH; j = jH; }
            a_min_est[i] = -a; a_max_est[i] = a; j_min_est[i] = -j; j_max_est[i] = j;
            continue;
        }
        /* ISO limits :157-171 */
        if (v_est[i] < 5.0) {
            a_min_est[i] = -5.0; a_max_est[i] = 4.0; j_min_est[i] = -5.0; j_max_est[i] = 5.0;
        } else if (v_est[i] < 20.0) {
            a_min_est[i] = -5.5 + v_est[i] / 10.0;
            a_max_est[i] = 14.0 / 3.0 - 2.0 * v_est[i] / 15.0;
            j_min_est[i] = -35.0 / 6.0 + v_est[i] / 6.0;
            j_max_est[i] = 35.0 / 6.0 - v_est[i] / 6.0;
        } else {
            a_min_est[i] = -3.5; a_max_est[i] = 2.0; j_min_est[i] = -2.5; j_max_est[i] = 2.5;
        }
    }
}

/* ABO/Functions/MPCs/LUTgearshift.m:17-41: gear ratio from the estimated speed.  The reference compares with strict
 * inequalities on both sides, so a speed exactly on a threshold leaves its output unassigned (a MATLAB error); here
 * such a speed takes the higher gear. */
double orc_lut_gearshift(const eepacc_vehicle* V, double v) {
    int g = 0;
    while (g < 7 && !(v < V->upSpd[g])) ++g;
    return V->tau_gb[g];
}

static int count_mb(const eepacc_settings* S) {
    int c = 0;
    if (S->Mb) for (int k = 0; k < S->N_hor; ++k) c += (S->Mb[k] == 1);
    return c;
}
int orc_ab_num_rows(const eepacc_settings* S) {
    return (S->ab_route_rows ? 18 : 14) * S->N_hor + 2 + count_mb(S);
}
int orc_fb_num_rows(const eepacc_settings* S) { return 26 * S->N_hor + 2 + 2 * count_mb(S); }

/* ABO/Functions/MPCs/CreateQP_AB.m:58-387, solverToUse == 1 branch */
void orc_create_qp_ab(const eepacc_settings* S, const eepacc_vehicle* V, double s_0, double v_0,
                      const double* s_est, const double* v_est, const double* s_tv_est, double t_0,
                      double a_minus1, double* H, double* c, double* G, double* g_lb, double* g_ub) {
    (void)s_0; (void)v_0;
    const int N = S->N_hor, n_x = 2, n_u = 5, n_x_u = n_x + n_u;
    const int nz = n_x_u * N + n_x;
    const int nC = orc_ab_num_rows(S);
    const double w_FC = S->ab_fuel_term ? S->W_AB[0] : 0.0;
    const double w_a = S->W_AB[1], w_j = S->W_AB[2], w_v = S->W_AB[3], w_h = S->W_AB[4],
                 w_s = S->W_AB[5], w_f = S->W_AB[6];
    double sl[EEPACC_MAX_HORIZON], v_lim[EEPACC_MAX_HORIZON], v_stop[EEPACC_MAX_HORIZON],
        v_TL[EEPACC_MAX_HORIZON], v_curv[EEPACC_MAX_HORIZON], a_min[EEPACC_MAX_HORIZON],
        a_max[EEPACC_MAX_HORIZON], j_min[EEPACC_MAX_HORIZON], j_max[EEPACC_MAX_HORIZON];
    orc_estimate_route_and_comfort_bounds(S, s_est, v_est, t_0, sl, v_lim, v_stop, v_TL, v_curv,
                                          a_min, a_max, j_min, j_max);          /* :51-52 */
    memset(H, 0, sizeof(double) * (size_t)nz * nz);
    memset(c, 0, sizeof(double) * nz);
    memset(G, 0, sizeof(double) * (size_t)nC * nz);
    const double s_min = 0.0, s_max = S->s_goal, v_min = 0.0, v_max = V->v_max;  /* :68-75 */
    /* 0-based offsets of the 1-based indices of :128-135 */
    const int scurr = 0, vcurr = 1, acurr = 2, aprev = 2 - n_x_u, xi_v = 3, xi_h = 4, xi_s = 5,
              xi_f = 6;
    int r = 0;
#define ROW_BEGIN() do { } while (0)
#define SETG(col, val) G[IDX(r, (col), nz)] = (val)
#define ROW_END(lo, hi) do { g_lb[r] = (lo); g_ub[r] = (hi); ++r; } while (0)
    for (int kk = 0; kk < N; ++kk) {
        const double T = S->Tvec[kk];
        const int o = kk * n_x_u;
        const double v_minInc = v_lim[kk] < v_curv[kk] ? v_lim[kk] : v_curv[kk];   /* :55 */
        if (S->ab_fuel_term == 2) {
            /* objective: fuel term, ICE map :154-159 (commented in the checked-in file; savedABMPCsolICEMAP.mat was
             * written with it), tau_est(k) = LUTgearshift(v_est(k)), EstimateRouteAndComfortBounds.m:63-66 */
            const double tau = orc_lut_gearshift(V, v_est[kk]);
            H[IDX(o + vcurr, o + vcurr, nz)] += w_FC * (2.0 * V->k01 * V->F2 * V->R_w / V->tau_fd / tau / V->eta_drive);
            c[o + vcurr] += w_FC * (V->k10 / V->R_w * V->tau_fd * tau);
            c[o + acurr] += w_FC * (V->k01 * V->lambda * V->m * V->R_w / V->tau_fd / tau / V->eta_drive);
        } else {
        /* objective: fuel term :162-166 */
        H[IDX(o + vcurr, o + vcurr, nz)] += w_FC * 2.0 * V->p01 * V->F2;
        c[o + vcurr] += w_FC * V->p10;
        c[o + acurr] += w_FC * V->p01 * V->lambda * V->m;
        }
        /* acceleration :169-170 */
        H[IDX(o + acurr, o + acurr, nz)] += 2.0 * w_a;
        /* jerk :173-180 */
        if (kk == 0) {
            H[IDX(o + acurr, o + acurr, nz)] += 2.0 * w_j / (T * T);
            c[o + acurr] -= 2.0 * w_j / T * a_minus1;
        } else {
            double q = 2.0 * w_j / (T * T);
            H[IDX(o + acurr, o + acurr, nz)] += q;
            H[IDX(o + acurr, o + aprev, nz)] -= q;
            H[IDX(o + aprev, o + acurr, nz)] -= q;
            H[IDX(o + aprev, o + aprev, nz)] += q;
        }
        /* slacks :183-187 */
        c[o + xi_v] += w_v;
        c[o + xi_h] += 1e2 * w_h;
        H[IDX(o + xi_h, o + xi_h, nz)] += 2.0 * w_h;
        c[o + xi_s] += w_s;
        c[o + xi_f] += w_f;
        /* bounds as rows :256-279 */
        SETG(o + scurr, 1.0); ROW_END(s_min, s_max);
        SETG(o + vcurr, 1.0); ROW_END(v_min, v_max);
        SETG(o + xi_v, 1.0);  ROW_END(0.0, INFINITY);
        SETG(o + xi_h, 1.0);  ROW_END(0.0, INFINITY);
        SETG(o + xi_s, 1.0);  ROW_END(0.0, INFINITY);
        SETG(o + xi_f, 1.0);  ROW_END(0.0, INFINITY);
        /* move blocking :283-289 */
        if (S->Mb && S->Mb[kk] == 1) {
            SETG(o + aprev, 1.0); SETG(o + acurr, -1.0); ROW_END(0.0, 0.0);
        }
        /* ISO acceleration :292-299 */
        SETG(o + acurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, a_max[kk]);
        SETG(o + acurr, 1.0); SETG(o + xi_f, 1.0);  ROW_END(a_min[kk], INFINITY);
        /* ISO jerk :302-322 */
        if (kk > 0) {
            SETG(o + aprev, -1.0); SETG(o + acurr, 1.0); SETG(o + xi_f, -1.0);
            ROW_END(-INFINITY, T * j_max[kk]);
            SETG(o + aprev, -1.0); SETG(o + acurr, 1.0); SETG(o + xi_f, 1.0);
            ROW_END(T * j_min[kk], INFINITY);
        } else {
            SETG(o + acurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, T * j_max[kk] + a_minus1);
            SETG(o + acurr, 1.0); SETG(o + xi_f, 1.0);  ROW_END(T * j_min[kk] + a_minus1, INFINITY);
        }
        /* speed / curve / stop / traffic-light caps: ORIG/.../CreateQP_AB.m:307-329
         * (commented out in ABO/.../CreateQP_AB.m:324-346) */
        if (S->ab_route_rows) {
            SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_lim[kk]);
            SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_curv[kk]);
            SETG(o + vcurr, 1.0); SETG(o + xi_s, -1.0); ROW_END(-INFINITY, v_stop[kk]);
            SETG(o + vcurr, 1.0); SETG(o + xi_s, -1.0); ROW_END(-INFINITY, v_TL[kk]);
        }
        /* minimum velocity incentive :349-352 */
        SETG(o + vcurr, 1.0); SETG(o + xi_v, 1.0); ROW_END(v_minInc, INFINITY);
        /* safe headway :355-362 */
        SETG(o + scurr, 1.0); SETG(o + xi_s, -1.0); ROW_END(-INFINITY, s_tv_est[kk] - S->h_min);
        SETG(o + scurr, 1.0); SETG(o + vcurr, S->tau_min); SETG(o + xi_s, -1.0);
        ROW_END(-INFINITY, s_tv_est[kk]);
        /* desired headway policy :365-371 */
        {
            const double T_hwp = 2.0, A_hwp = 2.0, G_hwp = -0.0246 * T_hwp + 0.010819;
            SETG(o + scurr, 1.0); SETG(o + vcurr, T_hwp + G_hwp * v_est[kk]); SETG(o + xi_h, -1.0);
            ROW_END(-INFINITY, s_tv_est[kk] - A_hwp);
        }
    }
    /* final stage :376-387 (uses s_tv_est(N), MATLAB 1-based) */
    {
        const int o = N * n_x_u;
        SETG(o + scurr, 1.0); ROW_END(-INFINITY, s_tv_est[N - 1] - S->h_min);
        SETG(o + scurr, 1.0); SETG(o + vcurr, S->tau_min); ROW_END(-INFINITY, s_tv_est[N - 1]);
    }
    if (r != nC) { fprintf(stderr, "orc_create_qp_ab: row count %d != %d\n", r, nC); abort(); }
}

/* ABO/Functions/MPCs/CreateQP_BL.m:36-334, solverToUse == 1 branch (dense qpOASES: no dynamics rows, z = [s v a xi_f]) */
int orc_bl_num_rows(const eepacc_settings* S) { return 13 * S->N_hor + 2; }

void orc_create_qp_bl(const eepacc_settings* S, const eepacc_vehicle* V, double s_0, double v_0,
                      const double* s_est, const double* v_est, const double* s_tv_est, double t_0,
                      double a_minus1, double* H, double* c, double* G, double* g_lb, double* g_ub) {
    (void)s_0; (void)v_0;
    const int N = S->N_hor, n_x = 2, n_u = 2, n_x_u = n_x + n_u;
    const int nz = n_x_u * N + n_x;
    const int nC = orc_bl_num_rows(S);
    const double w_v = S->W_BL[0], w_a = S->W_BL[1], w_j = S->W_BL[2], w_f = S->W_BL[3];   /* :36-39 */
    const double Ts = S->Tvec[0];                                                          /* :30 */
    double sl[EEPACC_MAX_HORIZON], v_lim[EEPACC_MAX_HORIZON], v_stop[EEPACC_MAX_HORIZON],
        v_TL[EEPACC_MAX_HORIZON], v_curv[EEPACC_MAX_HORIZON], a_min[EEPACC_MAX_HORIZON],
        a_max[EEPACC_MAX_HORIZON], j_min[EEPACC_MAX_HORIZON], j_max[EEPACC_MAX_HORIZON];
    orc_estimate_route_and_comfort_bounds(S, s_est, v_est, t_0, sl, v_lim, v_stop, v_TL, v_curv,
                                          a_min, a_max, j_min, j_max);          /* :44 (MPCtype 1) */
    memset(H, 0, sizeof(double) * (size_t)nz * nz);
    memset(c, 0, sizeof(double) * nz);
    memset(G, 0, sizeof(double) * (size_t)nC * nz);
    const double s_min = 0.0, s_max = S->s_goal, v_min = 0.0, v_max = V->v_max;  /* :58-62 */
    const int scurr = 0, vcurr = 1, acurr = 2, aprev = 2 - n_x_u, xi_f = 3;      /* :112-118, 0-based */
    int r = 0;
    for (int kk = 0; kk < N; ++kk) {
        const int o = kk * n_x_u;
        c[o + vcurr] -= w_v;                                                     /* :133 */
        H[IDX(o + acurr, o + acurr, nz)] += 2.0 * w_a;                           /* :136-137 */
        if (kk == 0) {                                                           /* :140-147 */
            H[IDX(o + acurr, o + acurr, nz)] += 2.0 * w_j / (Ts * Ts);
            c[o + acurr] -= 2.0 * w_j / Ts * a_minus1;
        } else {
            double q = 2.0 * w_j / (Ts * Ts);
            H[IDX(o + acurr, o + acurr, nz)] += q;
            H[IDX(o + acurr, o + aprev, nz)] -= q;
            H[IDX(o + aprev, o + acurr, nz)] -= q;
            H[IDX(o + aprev, o + aprev, nz)] += q;
        }
        c[o + xi_f] += w_f;                                                      /* :150 */
        /* bounds as rows :214-227 */
        SETG(o + scurr, 1.0); ROW_END(s_min, s_max);
        SETG(o + vcurr, 1.0); ROW_END(v_min, v_max);
        SETG(o + xi_f, 1.0);  ROW_END(0.0, INFINITY);
        /* acceleration :231-238 */
        SETG(o + acurr, 1.0); SETG(o + xi_f, 1.0);  ROW_END(a_min[kk], INFINITY);
        SETG(o + acurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, a_max[kk]);
        /* jerk :241-261 */
        if (kk > 0) {
            SETG(o + aprev, -1.0); SETG(o + acurr, 1.0); SETG(o + xi_f, 1.0);  ROW_END(Ts * j_min[kk], INFINITY);
            SETG(o + aprev, -1.0); SETG(o + acurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, Ts * j_max[kk]);
        } else {
            SETG(o + acurr, 1.0); SETG(o + xi_f, 1.0);  ROW_END(Ts * j_min[kk] + a_minus1, INFINITY);
            SETG(o + acurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, Ts * j_max[kk] + a_minus1);
        }
        /* speed limit, curve, stop, traffic light :264-288 */
        SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_lim[kk]);
        SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_curv[kk]);
        SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_stop[kk]);
        SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_TL[kk]);
        /* safe headway :291-298 */
        SETG(o + scurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, s_tv_est[kk] - S->h_min);
        SETG(o + scurr, 1.0); SETG(o + vcurr, S->tau_min); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, s_tv_est[kk]);
    }
    {   /* final stage :303-318 (s_tv_est(N), MATLAB 1-based) */
        const int o = N * n_x_u;
        c[o + vcurr] -= w_v;
        SETG(o + scurr, 1.0); ROW_END(-INFINITY, s_tv_est[N - 1] - S->h_min);
        SETG(o + scurr, 1.0); SETG(o + vcurr, S->tau_min); ROW_END(-INFINITY, s_tv_est[N - 1]);
    }
    if (r != nC) { fprintf(stderr, "orc_create_qp_bl: row count %d != %d\n", r, nC); abort(); }
}

/* ABO/Functions/MPCs/CreateQP_FB.m:158-489, solverToUse == 1 branch */
void orc_create_qp_fb(const eepacc_settings* S, const eepacc_vehicle* V, double s_0, double v_0,
                      const double* s_est, const double* v_est, const double* s_tv_est, double t_0,
                      double v_minus1, double a_minus1, double Fm_minus1, double Fb_minus1,
                      double* H, double* c, double* G, double* g_lb, double* g_ub,
                      double* theta_est) {
    (void)v_minus1; (void)Fm_minus1; (void)Fb_minus1;
    const int N = S->N_hor, n_x = 2, n_u = 6, n_x_u = n_x + n_u;
    const int nz = n_x_u * N + n_x;
    const int nC = orc_fb_num_rows(S);
    const double w_P = S->W_FB[0], w_a = S->W_FB[1], w_j = S->W_FB[2], w_v = S->W_FB[3],
                 w_h = S->W_FB[4], w_s = S->W_FB[5], w_f = S->W_FB[6];
    const double* b = S->b_quadr;
    double v_lim[EEPACC_MAX_HORIZON], v_stop[EEPACC_MAX_HORIZON], v_TL[EEPACC_MAX_HORIZON],
        v_curv[EEPACC_MAX_HORIZON], a_min[EEPACC_MAX_HORIZON], a_max[EEPACC_MAX_HORIZON],
        j_min[EEPACC_MAX_HORIZON], j_max[EEPACC_MAX_HORIZON];
    orc_estimate_route_and_comfort_bounds(S, s_est, v_est, t_0, theta_est, v_lim, v_stop, v_TL,
                                          v_curv, a_min, a_max, j_min, j_max);  /* :56-57 */
    memset(H, 0, sizeof(double) * (size_t)nz * nz);
    memset(c, 0, sizeof(double) * nz);
    memset(G, 0, sizeof(double) * (size_t)nC * nz);
    const double s_min = s_0, s_max = S->s_goal, v_min = 0.0, v_max = V->v_max;  /* :69-80 */
    const double Fm_min = -1e4, Fm_max = 1e4, Fb_min = -1e4, Fb_max = 0.0;
    const int scurr = 0, vcurr = 1, Fmcurr = 2, Fbcurr = 3, xi_v = 4, xi_h = 5, xi_s = 6, xi_f = 7;
    const int vprev = 1 - n_x_u, Fmprev = 2 - n_x_u, Fbprev = 3 - n_x_u;          /* :143-149 */
    const double lm = V->lambda * V->m, za = V->zeta_a;
    const double K = (30.0 / M_PI) * V->phi;
    double zeta_rg = 0.0, Dzeta_rg = 0.0;
    int r = 0;
    for (int kk = 0; kk < N; ++kk) {
        const double Tp = S->Tvec[kk];
        const int o = kk * n_x_u;
        const double th = theta_est[kk];
        const double v_minInc = v_lim[kk] < v_curv[kk] ? v_lim[kk] : v_curv[kk];   /* :60 */
        if (kk > 0) {                                                             /* :171-177 */
            double prev = zeta_rg;
            zeta_rg = V->m * V->g * (V->c_r * cos(th) + sin(th));
            Dzeta_rg = zeta_rg - prev;
        } else {
            zeta_rg = V->m * V->g * (V->c_r * cos(th) + sin(th));
        }
        /* power :181-184 */
        {
            int ii[2] = {o + vcurr, o + Fmcurr};
            double M[2][2] = {{2.0 * K * K * b[5], K * b[4]}, {K * b[4], 2.0 * b[3]}};
            for (int a = 0; a < 2; ++a)
                for (int bb = 0; bb < 2; ++bb) H[IDX(ii[a], ii[bb], nz)] += w_P * M[a][bb];
            c[ii[0]] += w_P * K * b[2];
            c[ii[1]] += w_P * b[1];
        }
        /* acceleration penalty :187-191 */
        {
            int ii[3] = {o + vcurr, o + Fmcurr, o + Fbcurr};
            double ve = v_est[kk];
            double M[3][3] = {{za * za * ve * ve + za * zeta_rg, -za * ve, -za * ve},
                              {-za * ve, 1.0, 1.0},
                              {-za * ve, 1.0, 1.0}};
            double f = 2.0 * w_a / (lm * lm);
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb) H[IDX(ii[a], ii[bb], nz)] += f * M[a][bb];
            double fc = w_a / (lm * lm);
            c[ii[1]] += fc * (-2.0 * zeta_rg);
            c[ii[2]] += fc * (-2.0 * zeta_rg);
        }
        /* jerk penalty :194-208 */
        if (kk == 0) {
            int ii[2] = {o + Fmcurr, o + Fbcurr};
            double f = 2.0 * w_j / ((lm * Tp) * (lm * Tp));
            for (int a = 0; a < 2; ++a)
                for (int bb = 0; bb < 2; ++bb) H[IDX(ii[a], ii[bb], nz)] += f;
            double th0 = theta_est[0];
            double cc = 2.0 * w_j *
                        (za * v_0 * v_0 + V->m * V->g * (V->c_r * cos(th0) + sin(th0)) + lm * a_minus1) /
                        ((lm * Tp) * (lm * Tp));
            c[ii[0]] -= cc;
            c[ii[1]] -= cc;
        } else {
            int ii[6] = {o + vprev, o + Fmprev, o + Fbprev, o + vcurr, o + Fmcurr, o + Fbcurr};
            double vk = v_est[kk], vp = v_est[kk - 1];
            double M[6][6] = {
                {za * za * vk * vk + 2.0 * za * Dzeta_rg, -za * vp, -za * vp, -za * za * vk * vp, za * vp, za * vp},
                {-za * vp, 1.0, 1.0, za * vk, -1.0, -1.0},
                {-za * vp, 1.0, 1.0, za * vk, -1.0, -1.0},
                {-za * za * vk * vp, za * vk, za * vk, za * za * vp * vp - 2.0 * za * Dzeta_rg, -za * vk, -za * vk},
                {za * vp, -1.0, -1.0, -za * vk, 1.0, 1.0},
                {za * vp, -1.0, -1.0, -za * vk, 1.0, 1.0}};
            double f = 2.0 * w_j / ((lm * Tp) * (lm * Tp));
            for (int a = 0; a < 6; ++a)
                for (int bb = 0; bb < 6; ++bb) H[IDX(ii[a], ii[bb], nz)] += f * M[a][bb];
            double e[6] = {0.0, 1.0, 1.0, 0.0, -1.0, -1.0};
            for (int a = 0; a < 6; ++a) c[ii[a]] += f * Dzeta_rg * e[a];
        }
        /* slacks :211-215 */
        c[o + xi_v] += w_v;
        c[o + xi_h] += 1e2 * w_h;
        H[IDX(o + xi_h, o + xi_h, nz)] += 2.0 * w_h;
        c[o + xi_s] += w_s;
        c[o + xi_f] += w_f;
        /* bounds as rows :311-342 */
        SETG(o + scurr, 1.0);  ROW_END(s_min, s_max);
        SETG(o + vcurr, 1.0);  ROW_END(v_min, v_max);
        SETG(o + Fmcurr, 1.0); ROW_END(Fm_min, Fm_max);
        SETG(o + Fbcurr, 1.0); ROW_END(Fb_min, Fb_max);
        SETG(o + xi_v, 1.0);   ROW_END(0.0, INFINITY);
        SETG(o + xi_h, 1.0);   ROW_END(0.0, INFINITY);
        SETG(o + xi_s, 1.0);   ROW_END(0.0, INFINITY);
        SETG(o + xi_f, 1.0);   ROW_END(0.0, INFINITY);
        /* move blocking :346-356 */
        if (S->Mb && S->Mb[kk] == 1) {
            SETG(o + Fmprev, 1.0); SETG(o + Fmcurr, -1.0); ROW_END(0.0, 0.0);
            SETG(o + Fbprev, 1.0); SETG(o + Fbcurr, -1.0); ROW_END(0.0, 0.0);
        }
        /* torque limits :359-366 */
        {
            double cv = V->phi * V->T_m_max * V->T_m_max / 4.0 / V->P_m_max;
            SETG(o + vcurr, -cv); SETG(o + Fmcurr, V->eta_TF / V->phi); SETG(o + xi_f, 1.0);
            ROW_END(-V->T_m_max, INFINITY);
            SETG(o + vcurr, cv); SETG(o + Fmcurr, 1.0 / V->eta_TF / V->phi); SETG(o + xi_f, -1.0);
            ROW_END(-INFINITY, V->T_m_max);
        }
        /* rear wheel traction :369-377 */
        {
            double zeta_w = V->m * V->g * (V->L_f * cos(th) + V->h_g * sin(th));
            SETG(o + Fmcurr, V->L / V->mu + V->h_g); SETG(o + Fbcurr, V->h_g); SETG(o + xi_f, 1.0);
            ROW_END(-zeta_w + V->h_g * zeta_rg, INFINITY);
            SETG(o + Fmcurr, V->L / V->mu - V->h_g); SETG(o + Fbcurr, -V->h_g); SETG(o + xi_f, -1.0);
            ROW_END(-INFINITY, zeta_w - V->h_g * zeta_rg);
        }
        /* total friction :380-387 */
        SETG(o + Fmcurr, 1.0); SETG(o + Fbcurr, 1.0); SETG(o + xi_f, -1.0);
        ROW_END(-INFINITY, V->mu * V->m * V->g * cos(th));
        SETG(o + Fmcurr, 1.0); SETG(o + Fbcurr, 1.0); SETG(o + xi_f, 1.0);
        ROW_END(-V->mu * V->m * V->g * cos(th), INFINITY);
        /* ISO acceleration :390-397 */
        {
            double base = za * v_est[kk] * v_est[kk] + zeta_rg;
            SETG(o + Fmcurr, 1.0); SETG(o + Fbcurr, 1.0); SETG(o + xi_f, -1.0);
            ROW_END(-INFINITY, lm * a_max[kk] + base);
            SETG(o + Fmcurr, 1.0); SETG(o + Fbcurr, 1.0); SETG(o + xi_f, 1.0);
            ROW_END(lm * a_min[kk] + base, INFINITY);
        }
        /* ISO jerk :400-418 */
        if (kk == 0) {
            double base = za * v_est[kk] * v_est[kk] + zeta_rg;
            SETG(o + Fmcurr, 1.0); SETG(o + Fbcurr, 1.0); SETG(o + xi_f, -1.0);
            ROW_END(-INFINITY, lm * (Tp * j_max[kk] + a_minus1) + base);
            SETG(o + Fmcurr, 1.0); SETG(o + Fbcurr, 1.0); SETG(o + xi_f, 1.0);
            ROW_END(lm * (Tp * j_min[kk] + a_minus1) + base, INFINITY);
        } else {
            double base = za * (v_est[kk] * v_est[kk] - v_est[kk - 1] * v_est[kk - 1]) + Dzeta_rg;
            SETG(o + Fmprev, -1.0); SETG(o + Fbprev, -1.0); SETG(o + Fmcurr, 1.0);
            SETG(o + Fbcurr, 1.0); SETG(o + xi_f, -1.0);
            ROW_END(-INFINITY, lm * Tp * j_max[kk] + base);
            SETG(o + Fmprev, -1.0); SETG(o + Fbprev, -1.0); SETG(o + Fmcurr, 1.0);
            SETG(o + Fbcurr, 1.0); SETG(o + xi_f, 1.0);
            ROW_END(lm * Tp * j_min[kk] + base, INFINITY);
        }
        /* speed / curve / stop / TL caps :421-442 */
        SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_lim[kk]);
        SETG(o + vcurr, 1.0); SETG(o + xi_f, -1.0); ROW_END(-INFINITY, v_curv[kk]);
        SETG(o + vcurr, 1.0); SETG(o + xi_s, -1.0); ROW_END(-INFINITY, v_stop[kk]);
        SETG(o + vcurr, 1.0); SETG(o + xi_s, -1.0); ROW_END(-INFINITY, v_TL[kk]);
        /* incentive :445-448 */
        SETG(o + vcurr, 1.0); SETG(o + xi_v, 1.0); ROW_END(v_minInc, INFINITY);
        /* safe headway :451-458 */
        SETG(o + scurr, 1.0); SETG(o + xi_s, -1.0); ROW_END(-INFINITY, s_tv_est[kk] - S->h_min);
        SETG(o + scurr, 1.0); SETG(o + vcurr, S->tau_min); SETG(o + xi_s, -1.0);
        ROW_END(-INFINITY, s_tv_est[kk]);
        /* headway policy :461-473 */
        {
            const double T_hwp = 2.0, A_hwp = 2.0, G_hwp = -0.0246 * T_hwp + 0.010819;
            if (S->FBuseTaylor) {
                SETG(o + scurr, 1.0); SETG(o + vcurr, T_hwp + 2.0 * G_hwp * v_est[kk]);
                SETG(o + xi_h, -1.0);
                ROW_END(-INFINITY, s_tv_est[kk] - A_hwp + G_hwp * v_est[kk] * v_est[kk]);
            } else {
                SETG(o + scurr, 1.0); SETG(o + vcurr, T_hwp + G_hwp * v_est[kk]);
                SETG(o + xi_h, -1.0);
                ROW_END(-INFINITY, s_tv_est[kk] - A_hwp);
            }
        }
    }
    {   /* final stage :478-489 */
        const int o = N * n_x_u;
        SETG(o + scurr, 1.0); ROW_END(-INFINITY, s_tv_est[N - 1] - S->h_min);
        SETG(o + scurr, 1.0); SETG(o + vcurr, S->tau_min); ROW_END(-INFINITY, s_tv_est[N - 1]);
    }
    if (r != nC) { fprintf(stderr, "orc_create_qp_fb: row count %d != %d\n", r, nC); abort(); }
}
#undef SETG
#undef ROW_END
#undef ROW_BEGIN

static void mat2_mul(const double* X, const double* Y, double* Z) { /* 2x2 row-major */
    double z0 = X[0] * Y[0] + X[1] * Y[2], z1 = X[0] * Y[1] + X[1] * Y[3];
    double z2 = X[2] * Y[0] + X[3] * Y[2], z3 = X[2] * Y[1] + X[3] * Y[3];
    Z[0] = z0; Z[1] = z1; Z[2] = z2; Z[3] = z3;
}

/* ABO/Functions/MPCs/TransformToDenseFormulation.m:30-91 */
void orc_transform_to_dense(int N, int nu, int nC, const double* A, const double* B,
                            const double* D, const double* Hs, const double* cs, const double* Gs,
                            const double* glb, const double* gub, double s_curr, double v_curr,
                            double* Hd, double* cd, double* Gd, double* lbd, double* ubd,
                            double* Psi, double* d) {
    const int nx = 2, nV = N * nu, nz = N * (nx + nu) + nx;
    memset(Psi, 0, sizeof(double) * (size_t)nz * nV);
    memset(d, 0, sizeof(double) * nz);
    const double x0[2] = {s_curr, v_curr};
    int o = 0;
    for (int k = 0; k <= N; ++k) {                                             /* :37 */
        if (k > 0) {
            for (int i = 0; i < k; ++i) {                                      /* :46-52 */
                double A_[4] = {1, 0, 0, 1};
                for (int j = 1; j <= k - i - 1; ++j) mat2_mul(A_, &A[4 * (k - j)], A_);
                const double* Bi = &B[(size_t)i * 2 * nu];
                for (int cc = 0; cc < nu; ++cc) {
                    Psi[IDX(o, i * nu + cc, nV)] = A_[0] * Bi[cc] + A_[1] * Bi[nu + cc];
                    Psi[IDX(o + 1, i * nu + cc, nV)] = A_[2] * Bi[cc] + A_[3] * Bi[nu + cc];
                }
            }
            {                                                                  /* :55-59 */
                double A_[4] = {1, 0, 0, 1};
                for (int i = 1; i <= k; ++i) mat2_mul(A_, &A[4 * (k - i)], A_);
                d[o] += A_[0] * x0[0] + A_[1] * x0[1];
                d[o + 1] += A_[2] * x0[0] + A_[3] * x0[1];
            }
            for (int i = 0; i < k; ++i) {                                      /* :62-68 */
                double A_[4] = {1, 0, 0, 1};
                for (int j = 1; j <= k - i - 1; ++j) mat2_mul(A_, &A[4 * (k - j)], A_);
                d[o] += A_[0] * D[2 * i] + A_[1] * D[2 * i + 1];
                d[o + 1] += A_[2] * D[2 * i] + A_[3] * D[2 * i + 1];
            }
        } else {
            d[o] = x0[0]; d[o + 1] = x0[1];                                    /* :72-73 */
        }
        o += nx;
        if (k < N) {                                                           /* :78-82 */
            for (int cc = 0; cc < nu; ++cc) Psi[IDX(o + cc, k * nu + cc, nV)] = 1.0;
            o += nu;
        }
    }
    /* literal dense products :87-91 */
    double* T1 = (double*)calloc((size_t)nz * nV, sizeof(double));   /* Hs * Psi */
    for (int i = 0; i < nz; ++i)
        for (int k = 0; k < nz; ++k) {
            double h = Hs[IDX(i, k, nz)];
            if (h == 0.0) continue;   /* value-identical shortcut: adding h*Psi with h == 0 */
            const double* pk = &Psi[IDX(k, 0, nV)];
            double* t = &T1[IDX(i, 0, nV)];
            for (int j = 0; j < nV; ++j) t[j] += h * pk[j];
        }
    memset(Hd, 0, sizeof(double) * (size_t)nV * nV);
    for (int k = 0; k < nz; ++k) {
        const double* pk = &Psi[IDX(k, 0, nV)];
        const double* tk = &T1[IDX(k, 0, nV)];
        for (int i = 0; i < nV; ++i) {
            double p = pk[i];
            if (p == 0.0) continue;
            double* h = &Hd[IDX(i, 0, nV)];
            for (int j = 0; j < nV; ++j) h[j] += p * tk[j];
        }
    }
    double* tmp = (double*)calloc(nz, sizeof(double));               /* .5(Hs+Hs')d + cs */
    for (int i = 0; i < nz; ++i) {
        double s = 0.0;
        for (int k = 0; k < nz; ++k) s += 0.5 * (Hs[IDX(i, k, nz)] + Hs[IDX(k, i, nz)]) * d[k];
        tmp[i] = s + cs[i];
    }
    for (int j = 0; j < nV; ++j) cd[j] = 0.0;
    for (int k = 0; k < nz; ++k) {
        const double* pk = &Psi[IDX(k, 0, nV)];
        for (int j = 0; j < nV; ++j) cd[j] += pk[j] * tmp[k];
    }
    memset(Gd, 0, sizeof(double) * (size_t)nC * nV);
    for (int i = 0; i < nC; ++i) {
        double gd = 0.0;
        double* g = &Gd[IDX(i, 0, nV)];
        for (int k = 0; k < nz; ++k) {
            double gv = Gs[IDX(i, k, nz)];
            if (gv == 0.0) continue;
            gd += gv * d[k];
            const double* pk = &Psi[IDX(k, 0, nV)];
            for (int j = 0; j < nV; ++j) g[j] += gv * pk[j];
        }
        lbd[i] = glb[i] - gd;
        ubd[i] = gub[i] - gd;
    }
    free(T1); free(tmp);
}

/* ABO/Functions/MPCs/RunPlantModel.m:27-44 */
static void f_NL(const eepacc_settings* S, const eepacc_vehicle* V, const double x[2], double u,
                 double xd[2]) {
    double theta = orc_interp_pwa(x[0], S->s_slope, S->slope, S->n_slope);
    xd[0] = x[1];
    xd[1] = 1.0 / V->lambda / V->m *
            (u - V->zeta_a * x[1] * x[1] - V->c_r * V->m * V->g * cos(theta) - V->m * V->g * sin(theta));
}
void orc_run_plant_model(const eepacc_settings* S, const eepacc_vehicle* V, double s, double v,
                         double Fm, double Fb, double* s_next, double* v_next) {
    const int M = S->N_integratePlant;
    const double Ts = S->Tvec[0], u = Fm + Fb, DT = Ts / M;
    double x[2] = {s, v};
    for (int k = 0; k < M; ++k) {
        double k1[2], k2[2], k3[2], k4[2], t[2];
        f_NL(S, V, x, u, k1);
        t[0] = x[0] + DT / 2 * k1[0]; t[1] = x[1] + DT / 2 * k1[1];
        f_NL(S, V, t, u, k2);
        t[0] = x[0] + DT / 2 * k2[0]; t[1] = x[1] + DT / 2 * k2[1];
        f_NL(S, V, t, u, k3);
        t[0] = x[0] + DT * k3[0]; t[1] = x[1] + DT * k3[1];
        f_NL(S, V, t, u, k4);
        x[0] = x[0] + DT / 6 * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0]);
        x[1] = x[1] + DT / 6 * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1]);
    }
    *s_next = x[0];
    *v_next = x[1];
}

/* force allocation: ABO/RunOpt_ABMPC.m:287-324 */
static void ab_force_allocation(const eepacc_settings* S, const eepacc_vehicle* V, double s_meas,
                                double v_meas, double a_qp, double* Fm, double* Fb, double* a_real) {
    double theta = orc_interp_pwa(s_meas, S->s_slope, S->slope, S->n_slope);
    double F_r = -V->zeta_a * v_meas * v_meas - V->c_r * V->m * V->g * cos(theta) - V->m * V->g * sin(theta);
    double F_t_req = V->m * V->lambda * a_qp - F_r;
    double F_f_r_max = V->mu / V->L *
                       (V->m * V->g * (V->L_f * cos(theta) + V->h_g * sin(theta)) +
                        V->h_g * (V->zeta_a * v_meas * v_meas + V->lambda * V->m * a_qp));
    double F_f_tot_max = V->mu * V->m * V->g * cos(theta);
    if (F_t_req < 0.0) {
        double F_m_min = (v_meas < V->omega_m_r / V->phi) ? -V->phi * V->T_m_max / V->eta_TF
                                                          : -V->P_m_max / V->eta_TF / v_meas;
        double fm = F_t_req;
        if (F_m_min > fm) fm = F_m_min;
        if (-F_f_r_max > fm) fm = -F_f_r_max;
        *Fm = fm;
        *Fb = (F_t_req > -F_f_tot_max ? F_t_req : -F_f_tot_max) - fm;
    } else {
        double F_m_max = (v_meas < V->omega_m_r / V->phi) ? V->phi * V->T_m_max * V->eta_TF
                                                          : V->P_m_max * V->eta_TF / v_meas;
        double fm = F_t_req;
        if (F_m_max < fm) fm = F_m_max;
        if (F_f_r_max < fm) fm = F_f_r_max;
        *Fm = fm;
        *Fb = 0.0;
    }
    *a_real = (*Fm + *Fb + F_r) / V->m / V->lambda;
}

/* One ABMPC step: ABO/RunOpt_ABMPC.m:193-329 */
int orc_ab_step(const eepacc_settings* S, const eepacc_vehicle* V, orc_step_io* io,
                double* dense_out) {
    /* bl_mode: the same loop body is ABO/RunOpt_BLMPC.m:175-300 with n_u = 2 and CreateQP_BL */
    const int bl = S->bl_mode != 0;
    const int N = S->N_hor, nu = bl ? 2 : 5, nx = 2;
    const int nz = N * (nx + nu) + nx, nV = N * nu, nC = bl ? orc_bl_num_rows(S) : orc_ab_num_rows(S);
    double s_est[EEPACC_MAX_HORIZON + 1], v_est[EEPACC_MAX_HORIZON + 1];
    double s_tv_est[EEPACC_MAX_HORIZON + 1], v_tv_est[EEPACC_MAX_HORIZON + 1];
    /* :194,197 (paramEstSetting 2 would need the previous solution: io->s_pred/v_pred in) */
    orc_estimate_vehicle_trajectory(S, 0, io->s, io->v, io->a_prev, io->s_pred, io->v_pred, s_est, v_est);
    orc_estimate_vehicle_trajectory(S, 1, io->s_tv, io->v_tv, io->a_tv_prev, NULL, NULL, s_tv_est, v_tv_est);
    double* Hs = (double*)malloc(sizeof(double) * (size_t)nz * nz);
    double* cs = (double*)malloc(sizeof(double) * nz);
    double* Gs = (double*)malloc(sizeof(double) * (size_t)nC * nz);
    double* glb = (double*)malloc(sizeof(double) * nC);
    double* gub = (double*)malloc(sizeof(double) * nC);
    double* Hd = (double*)malloc(sizeof(double) * (size_t)nV * nV);
    double* cd = (double*)malloc(sizeof(double) * nV);
    double* Gd = (double*)malloc(sizeof(double) * (size_t)nC * nV);
    double* lbd = (double*)malloc(sizeof(double) * nC);
    double* ubd = (double*)malloc(sizeof(double) * nC);
    double* Psi = (double*)malloc(sizeof(double) * (size_t)nz * nV);
    double* d = (double*)malloc(sizeof(double) * nz);
    double* x = (double*)calloc(nV, sizeof(double));
    double* z = (double*)malloc(sizeof(double) * nz);
    double A[EEPACC_MAX_HORIZON * 4], Bm[EEPACC_MAX_HORIZON * 2 * 5], Dm[EEPACC_MAX_HORIZON * 2];
    if (bl) orc_create_qp_bl(S, V, io->s, io->v, s_est, v_est, s_tv_est, io->t0, io->a_prev, Hs, cs, Gs, glb, gub);
    else orc_create_qp_ab(S, V, io->s, io->v, s_est, v_est, s_tv_est, io->t0, io->a_prev, Hs, cs, Gs, glb, gub); /* :204 */
    memset(Bm, 0, sizeof(Bm));
    for (int k = 0; k < N; ++k) {                                            /* :74-82 */
        double T = S->Tvec[k];
        A[4 * k] = 1; A[4 * k + 1] = T; A[4 * k + 2] = 0; A[4 * k + 3] = 1;
        Bm[(size_t)k * 2 * nu] = 0.5 * T * T;
        Bm[(size_t)k * 2 * nu + nu] = T;
        Dm[2 * k] = 0; Dm[2 * k + 1] = 0;
    }
    orc_transform_to_dense(N, nu, nC, A, Bm, Dm, Hs, cs, Gs, glb, gub, io->s, io->v, Hd, cd, Gd,
                           lbd, ubd, Psi, d);                                 /* :238 */
    double cost = 0.0;
    double bl_eps = 0.0;
    if (bl && S->W_BL[1] == 0.0 && S->W_BL[2] == 0.0) {
        /* the reference's baseline weights make this a linear program (H = 0), which qpOASES regularises internally.
         * Here: curvature eps on the accelerations (the least-norm LP optimum for eps below a data-dependent
         * threshold); the slack columns keep the solver's own relative floor.  eps is taken out of the cost again. */
        bl_eps = S->bl_lp_eps > 0.0 ? S->bl_lp_eps : 1e-4;
        for (int k = 0; k < N; ++k) Hd[IDX(k * nu, k * nu, nV)] += bl_eps;
    }
    orc_qp_solve_dense(nV, nC, Hd, cd, Gd, lbd, ubd, NULL, NULL, NULL, 0.0, 0, x, &cost, &io->qp); /* :252 */
    if (bl_eps > 0.0) {
        for (int k = 0; k < N; ++k) { cost -= 0.5 * bl_eps * x[k * nu] * x[k * nu]; Hd[IDX(k * nu, k * nu, nV)] -= bl_eps; }
    }
    for (int i = 0; i < nz; ++i) {                                           /* :261 */
        double sacc = d[i];
        for (int j = 0; j < nV; ++j) sacc += Psi[IDX(i, j, nV)] * x[j];
        z[i] = sacc;
    }
    for (int k = 0; k <= N; ++k) { io->s_pred[k] = z[k * (nx + nu)]; io->v_pred[k] = z[k * (nx + nu) + 1]; } /* :284-285 */
    double Fm, Fb, a_real;
    ab_force_allocation(S, V, io->s, io->v, z[2], &Fm, &Fb, &a_real);          /* :287-324 */
    io->out[EEPACC_OUT_S] = z[0];
    io->out[EEPACC_OUT_V] = z[1];
    io->out[EEPACC_OUT_FM] = Fm;
    io->out[EEPACC_OUT_FB] = Fb;
    io->out[EEPACC_OUT_A] = a_real;
    io->out[EEPACC_OUT_XI_V] = bl ? 0.0 : z[3];
    io->out[EEPACC_OUT_XI_H] = bl ? 0.0 : z[4];
    io->out[EEPACC_OUT_XI_S] = bl ? 0.0 : z[5];
    io->out[EEPACC_OUT_XI_F] = bl ? z[3] : z[6];                               /* BL :252 */
    io->out[EEPACC_OUT_COST] = cost;
    io->out[EEPACC_OUT_DISTHOR] = s_est[N] - io->s;                           /* :200 */
    io->out[EEPACC_OUT_AQP] = z[2];
    if (dense_out) {
        double* p = dense_out;
        memcpy(p, Hd, sizeof(double) * (size_t)nV * nV); p += (size_t)nV * nV;
        memcpy(p, cd, sizeof(double) * nV); p += nV;
        memcpy(p, Gd, sizeof(double) * (size_t)nC * nV); p += (size_t)nC * nV;
        memcpy(p, lbd, sizeof(double) * nC); p += nC;
        memcpy(p, ubd, sizeof(double) * nC); p += nC;
        memcpy(p, x, sizeof(double) * nV);
    }
    int st = io->qp.status;
    free(Hs); free(cs); free(Gs); free(glb); free(gub); free(Hd); free(cd); free(Gd);
    free(lbd); free(ubd); free(Psi); free(d); free(x); free(z);
    return st;
}

/* One FBMPC step: ABO/RunOpt_FBMPC.m:204-320 */
int orc_fb_step(const eepacc_settings* S, const eepacc_vehicle* V, orc_loop_state* st,
                orc_step_io* io, double* dense_out) {
    const int N = S->N_hor, nu = 6, nx = 2;
    const int nz = N * (nx + nu) + nx, nV = N * nu, nC = orc_fb_num_rows(S);
    double s_est[EEPACC_MAX_HORIZON + 1], v_est[EEPACC_MAX_HORIZON + 1];
    double s_tv_est[EEPACC_MAX_HORIZON + 1], v_tv_est[EEPACC_MAX_HORIZON + 1];
    double theta_est[EEPACC_MAX_HORIZON];
    orc_estimate_vehicle_trajectory(S, 0, io->s, io->v, io->a_prev, io->s_pred, io->v_pred, s_est, v_est);
    orc_estimate_vehicle_trajectory(S, 1, io->s_tv, io->v_tv, io->a_tv_prev, NULL, NULL, s_tv_est, v_tv_est);
    double* Hs = (double*)malloc(sizeof(double) * (size_t)nz * nz);
    double* cs = (double*)malloc(sizeof(double) * nz);
    double* Gs = (double*)malloc(sizeof(double) * (size_t)nC * nz);
    double* glb = (double*)malloc(sizeof(double) * nC);
    double* gub = (double*)malloc(sizeof(double) * nC);
    double* Hd = (double*)malloc(sizeof(double) * (size_t)nV * nV);
    double* cd = (double*)malloc(sizeof(double) * nV);
    double* Gd = (double*)malloc(sizeof(double) * (size_t)nC * nV);
    double* lbd = (double*)malloc(sizeof(double) * nC);
    double* ubd = (double*)malloc(sizeof(double) * nC);
    double* Psi = (double*)malloc(sizeof(double) * (size_t)nz * nV);
    double* d = (double*)malloc(sizeof(double) * nz);
    double* x = (double*)calloc(nV, sizeof(double));
    double* z = (double*)malloc(sizeof(double) * nz);
    double A[EEPACC_MAX_HORIZON * 4], Bm[EEPACC_MAX_HORIZON * 2 * 6], Dm[EEPACC_MAX_HORIZON * 2];
    orc_create_qp_fb(S, V, io->s, io->v, s_est, v_est, s_tv_est, io->t0, io->v_prev, io->a_prev,
                     io->Fm_prev, io->Fb_prev, Hs, cs, Gs, glb, gub, theta_est);   /* :215 */
    const double lm = V->lambda * V->m;
    /* relinearisation with the A(k,...) / D(k,:) index quirk, ABO/RunOpt_FBMPC.m:247-259:
     * the loop over i writes row k (= MPC step index, 1-based) every time, so after the loop
     * row k holds the i = N_hor values; rows > N_hor are never read. */
    if (S->FBuseTaylor) {
        if (st->k < N) {
            int i = N - 1;
            st->fbA22[st->k] = 1.0 - 2.0 * S->Tvec[i] * V->zeta_a * v_est[i] / lm;
            st->fbD2[st->k] = S->Tvec[i] / lm *
                              (V->zeta_a * v_est[i] * v_est[i] -
                               V->m * V->g * (V->c_r * cos(theta_est[i]) + sin(theta_est[i])));
        }
    } else {
        for (int i = 0; i < N; ++i)                                           /* :256 */
            st->fbD2[i] = S->Tvec[i] / lm *
                          (-V->zeta_a * v_est[i] * v_est[i] -
                           V->m * V->g * (V->c_r * cos(theta_est[i]) + sin(theta_est[i])));
    }
    memset(Bm, 0, sizeof(Bm));
    for (int k = 0; k < N; ++k) {
        double T = S->Tvec[k];
        /* A(k,1,2): the quirk writes Tvec(i=N_hor) into row k; identical for constant Tvec */
        A[4 * k] = 1; A[4 * k + 1] = T; A[4 * k + 2] = 0; A[4 * k + 3] = st->fbA22[k];
        Bm[(size_t)k * 2 * nu + nu] = T / lm;                                 /* :88 */
        Bm[(size_t)k * 2 * nu + nu + 1] = T / lm;
        Dm[2 * k] = 0; Dm[2 * k + 1] = st->fbD2[k];
    }
    orc_transform_to_dense(N, nu, nC, A, Bm, Dm, Hs, cs, Gs, glb, gub, io->s, io->v, Hd, cd, Gd,
                           lbd, ubd, Psi, d);                                 /* :264 */
    double cost = 0.0;
    /* the FB dense Hessian is indefinite: spectral regularisation (qp_dense.c) */
    orc_qp_solve_dense(nV, nC, Hd, cd, Gd, lbd, ubd, NULL, NULL, st->xwarm, -1e-8, 0, x, &cost, &io->qp); /* :278 */
    /* The FB QP is non-convex (bilinear speed x motor-force power term): it can have several KKT points.  Where
     * the predicted speed is zero the split of the total force into motor and friction brake costs nothing, and a
     * proximal sequence that passed through a braking phase can settle on a point of that flat face with the
     * friction brake still applied, which then pins the speed at zero (a local solution with a higher objective).
     * The reference's saved solutions never use the friction brake (Fb_opt = 0 for k >= 1), so the solve is
     * repeated from the same point with the friction brake released and the solution with the lower objective
     * is kept. */
    {
        int braking = 0;
        for (int k = 0; k < N; ++k) if (x[k * nu + 1] < -1e-6) braking = 1;
        if (braking) {
            double* x0b = (double*)malloc(sizeof(double) * nV);
            double* xb = (double*)calloc(nV, sizeof(double));
            double costb = 0.0;
            orc_qp_stats qb;
            memcpy(x0b, x, sizeof(double) * nV);
            for (int k = 0; k < N; ++k) { x0b[k * nu] += x0b[k * nu + 1]; x0b[k * nu + 1] = 0.0; }
            orc_qp_solve_dense(nV, nC, Hd, cd, Gd, lbd, ubd, NULL, NULL, x0b, -1e-8, 0, xb, &costb, &qb);
            if (qb.status == 0 && (io->qp.status != 0 || costb < cost - 1e-12 * fabs(cost))) {
                memcpy(x, xb, sizeof(double) * nV);
                cost = costb;
                qb.iterations += io->qp.iterations;
                io->qp = qb;
            }
            free(x0b); free(xb);
        }
    }
    memcpy(st->xwarm, x, sizeof(double) * nV);
    for (int i = 0; i < nz; ++i) {
        double sacc = d[i];
        for (int j = 0; j < nV; ++j) sacc += Psi[IDX(i, j, nV)] * x[j];
        z[i] = sacc;
    }
    for (int k = 0; k <= N; ++k) { io->s_pred[k] = z[k * (nx + nu)]; io->v_pred[k] = z[k * (nx + nu) + 1]; }
    io->out[EEPACC_OUT_S] = z[0];
    io->out[EEPACC_OUT_V] = z[1];
    io->out[EEPACC_OUT_FM] = z[2];                                            /* :294-299 */
    io->out[EEPACC_OUT_FB] = z[3];
    io->out[EEPACC_OUT_A] = 0.0;        /* filled by the loop driver, :316-318 */
    io->out[EEPACC_OUT_XI_V] = z[4];
    io->out[EEPACC_OUT_XI_H] = z[5];
    io->out[EEPACC_OUT_XI_S] = z[6];
    io->out[EEPACC_OUT_XI_F] = z[7];
    io->out[EEPACC_OUT_COST] = cost;
    io->out[EEPACC_OUT_DISTHOR] = s_est[N] - io->s;
    io->out[EEPACC_OUT_AQP] = 0.0;
    if (dense_out) {
        double* p = dense_out;
        memcpy(p, Hd, sizeof(double) * (size_t)nV * nV); p += (size_t)nV * nV;
        memcpy(p, cd, sizeof(double) * nV); p += nV;
        memcpy(p, Gd, sizeof(double) * (size_t)nC * nV); p += (size_t)nC * nV;
        memcpy(p, lbd, sizeof(double) * nC); p += nC;
        memcpy(p, ubd, sizeof(double) * nC); p += nC;
        memcpy(p, x, sizeof(double) * nV);
    }
    int stt = io->qp.status;
    free(Hs); free(cs); free(Gs); free(glb); free(gub); free(Hd); free(cd); free(Gd);
    free(lbd); free(ubd); free(Psi); free(d); free(x); free(z);
    return stt;
}

/* Closed loop: ABO/RunOpt_ABMPC.m:154-340 */
int orc_run_abmpc(const eepacc_settings* S, const eepacc_vehicle* V, int n_steps, double s0,
                  double v0, double a_minus1, const double* s_tv, const double* v_tv,
                  double* traj, int* status, int* qp_iters) {
    const double Ts = S->Tvec[0];
    orc_step_io io;
    memset(&io, 0, sizeof(io));
    double t_0 = 0.0, v_tv_measured = 0.0;
    int bad = 0;
    for (int kk = 0; kk < n_steps; ++kk) {
        if (kk == 0) {                                                        /* :159-172 */
            io.s = s0; io.v = v0; io.a_prev = a_minus1;
            memset(io.s_pred, 0, sizeof(io.s_pred));
            memset(io.v_pred, 0, sizeof(io.v_pred));
            io.s_tv = s_tv[0];
            v_tv_measured = 0.0;
            io.v_tv = 0.0;
            io.a_tv_prev = 0.0;
        } else {                                                              /* :173-191 */
            const double* prev = &traj[(size_t)(kk - 1) * EEPACC_OUT_N];
            double s_prev = prev[EEPACC_OUT_S], v_prev = prev[EEPACC_OUT_V];
            double s_m, v_m;
            orc_run_plant_model(S, V, s_prev, v_prev, prev[EEPACC_OUT_FM], prev[EEPACC_OUT_FB], &s_m, &v_m);
            io.s = s_m; io.v = v_m;
            io.a_prev = (v_m - v_prev) / Ts;
            io.s_tv = s_tv[kk];
            double v_tv_prev = v_tv_measured;
            v_tv_measured = v_tv[kk];
            io.v_tv = v_tv_measured;
            io.a_tv_prev = (v_tv_measured - v_tv_prev) / Ts;
        }
        io.t0 = t_0;
        int st = orc_ab_step(S, V, &io, NULL);
        memcpy(&traj[(size_t)kk * EEPACC_OUT_N], io.out, sizeof(double) * EEPACC_OUT_N);
        if (status) status[kk] = st;
        if (qp_iters) qp_iters[kk] = io.qp.iterations;
        bad += (st != 0);
        t_0 += Ts;                                                            /* :329 */
    }
    return bad;
}

/* Closed loop: ABO/RunOpt_FBMPC.m:161-331 */
int orc_run_fbmpc(const eepacc_settings* S, const eepacc_vehicle* V, int n_steps, double s0,
                  double v0, double a_minus1, const double* s_tv, const double* v_tv,
                  double* traj, int* status, int* qp_iters) {
    const double Ts = S->Tvec[0];
    const int N = S->N_hor;
    orc_step_io io;
    orc_loop_state st;
    memset(&io, 0, sizeof(io));
    memset(&st, 0, sizeof(st));
    const double lm = V->lambda * V->m;
    for (int k = 0; k < N; ++k) {                                             /* :78-90 */
        if (S->FBuseTaylor) {
            st.fbA22[k] = 1.0 - 2.0 * S->Tvec[k] * V->zeta_a * v0 * v0 / lm;
            st.fbD2[k] = S->Tvec[k] / lm * (V->zeta_a * v0 * v0);
        } else {
            st.fbA22[k] = 1.0;
            st.fbD2[k] = S->Tvec[k] / lm * (-V->zeta_a * v0 * v0);
        }
    }
    double t_0 = 0.0, v_tv_measured = 0.0;
    int bad = 0;
    for (int kk = 0; kk < n_steps; ++kk) {
        st.k = kk;
        if (kk == 0) {
            io.s = s0; io.v = v0; io.a_prev = a_minus1;
            io.Fm_prev = 0.0; io.Fb_prev = 0.0; io.v_prev = 5.0;              /* :62, :176-177 */
            io.s_tv = s_tv[0]; io.v_tv = 0.0; io.a_tv_prev = 0.0;
            v_tv_measured = 0.0;
        } else {
            const double* prev = &traj[(size_t)(kk - 1) * EEPACC_OUT_N];
            double s_prev = prev[EEPACC_OUT_S], v_prev = prev[EEPACC_OUT_V];
            double s_m, v_m;
            orc_run_plant_model(S, V, s_prev, v_prev, prev[EEPACC_OUT_FM], prev[EEPACC_OUT_FB], &s_m, &v_m);
            io.s = s_m; io.v = v_m; io.v_prev = v_prev;
            io.Fm_prev = prev[EEPACC_OUT_FM]; io.Fb_prev = prev[EEPACC_OUT_FB];
            io.a_prev = (v_m - v_prev) / Ts;
            io.s_tv = s_tv[kk];
            double v_tv_prev = v_tv_measured;
            v_tv_measured = v_tv[kk];
            io.v_tv = v_tv_measured;
            io.a_tv_prev = (v_tv_measured - v_tv_prev) / Ts;
        }
        io.t0 = t_0;
        int rc = orc_fb_step(S, V, &st, &io, NULL);
        if (kk > 0)                                                           /* :316-318 */
            io.out[EEPACC_OUT_A] = (io.out[EEPACC_OUT_V] - traj[(size_t)(kk - 1) * EEPACC_OUT_N + EEPACC_OUT_V]) / Ts;
        memcpy(&traj[(size_t)kk * EEPACC_OUT_N], io.out, sizeof(double) * EEPACC_OUT_N);
        if (status) status[kk] = rc;
        if (qp_iters) qp_iters[kk] = io.qp.iterations;
        bad += (rc != 0);
        t_0 += Ts;
    }
    return bad;
}

/* ABO/RunOpt_ABMPC.m:343-349 */
void orc_postprocess(const eepacc_settings* S, const eepacc_vehicle* V, int n, const double* v,
                     const double* Fm, double* rpm, double* Tm, double* P, double* E) {
    const double Ts = S->Tvec[0];
    double acc = 0.0;
    for (int k = 0; k < n; ++k) {
        rpm[k] = (30.0 / M_PI) * v[k] * V->phi;
        double sg = (Fm[k] > 0.0) - (Fm[k] < 0.0);
        Tm[k] = Fm[k] / V->phi / pow(V->eta_TF, sg);
        P[k] = orc_motor_power_fifth(Fm[k], rpm[k], S->b_fifthOrder);
        acc += P[k];
        E[k] = Ts * acc;
    }
}
