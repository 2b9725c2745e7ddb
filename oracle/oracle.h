/*
 * oracle.h -- TEST INFRASTRUCTURE.  CPU fp64 restatement of the reference's per-step
 * RunOpt_ABMPC / RunOpt_FBMPC pipeline.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libeepacc, HIP) never links or calls it.
 *
 * Paths: ABO/ = /root/reference/ACCMPC-ABO_CasADi/, ORIG/ = /root/reference/MATLAB_CasADi/.
 */
#ifndef EEPACC_ORACLE_H
#define EEPACC_ORACLE_H

#include "eepacc.h"   /* eepacc_settings / eepacc_vehicle PODs and EEPACC_OUT_* layout */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_qp_stats {
    int status;            /* 0 solved, 1 failed (mirrors ~QPsolver.stats().success)      */
    int iterations;        /* working-set changes over all proximal rounds                 */
    int prox_iterations;
    int n_active;
    int polished;          /* 1: exact KKT solve on the final working set verified         */
    double kkt_stationarity, kkt_primal, kkt_dual, rho;
} orc_qp_stats;

/* conic('qpoases') stand-in: ABO/RunOpt_ABMPC.m:252.  Row-major H[nV][nV], A[nC][nV]. */
int orc_qp_solve_dense(int nV, int nC, const double* H, const double* g, const double* A,
                       const double* lba, const double* uba, const double* lbx, const double* ubx,
                       const double* x0, double rho_rel, int max_prox,
                       double* x, double* cost, orc_qp_stats* st);

/* A2: ABO/Functions/MPCs/EstimateVehicleTrajectory.m:55-88. estSetting 0 ego / 1 lead. */
void orc_estimate_vehicle_trajectory(const eepacc_settings* S, int estSetting, double s_curr,
                                     double v_curr, double a_curr, const double* s_prev_sol,
                                     const double* v_prev_sol, double* s_est, double* v_est);

/* A3: ABO/Functions/MPCs/EstimateRouteAndComfortBounds.m:63-208 (MPCtype 0). Outputs [N]. */
void orc_estimate_route_and_comfort_bounds(const eepacc_settings* S, const double* s_est,
                                           const double* v_est, double t_0, double* slope_est,
                                           double* v_lim_max, double* v_stop_max, double* v_TL_max,
                                           double* v_curv_max, double* a_min_est, double* a_max_est,
                                           double* j_min_est, double* j_max_est);

/* ABO/Functions/MPCs/LUTgearshift.m:17-41 */
double orc_lut_gearshift(const eepacc_vehicle* V, double v);

/* number of constraint rows of the trimmed sparse-form QP / variables */
int orc_ab_num_rows(const eepacc_settings* S);
int orc_bl_num_rows(const eepacc_settings* S);      /* ABO/Functions/MPCs/CreateQP_BL.m, dense form: 13 N + 2 */
int orc_fb_num_rows(const eepacc_settings* S);

/* A4: ABO/Functions/MPCs/CreateQP_AB.m:58-387 (solverToUse == 1), rows already trimmed as
 * ABO/RunOpt_ABMPC.m:229-233 does.  H[nz][nz], c[nz], G[nC][nz], row-major, nz = 7N+2. */
/* ABO/Functions/MPCs/CreateQP_BL.m:36-334 (baseline controller; S->bl_mode = 1 makes orc_ab_step / orc_run_abmpc the
 * loop of ABO/RunOpt_BLMPC.m) */
void orc_create_qp_bl(const eepacc_settings* S, const eepacc_vehicle* V, double s_0, double v_0,
                      const double* s_est, const double* v_est, const double* s_tv_est, double t_0,
                      double a_minus1, double* H, double* c, double* G, double* g_lb, double* g_ub);
void orc_create_qp_ab(const eepacc_settings* S, const eepacc_vehicle* V, double s_0, double v_0,
                      const double* s_est, const double* v_est, const double* s_tv_est, double t_0,
                      double a_minus1, double* H, double* c, double* G, double* g_lb, double* g_ub);

/* F1: ABO/Functions/MPCs/CreateQP_FB.m:158-489. nz = 8N+2. theta_est[N] out. */
void orc_create_qp_fb(const eepacc_settings* S, const eepacc_vehicle* V, double s_0, double v_0,
                      const double* s_est, const double* v_est, const double* s_tv_est, double t_0,
                      double v_minus1, double a_minus1, double Fm_minus1, double Fb_minus1,
                      double* H, double* c, double* G, double* g_lb, double* g_ub,
                      double* theta_est);

/* A5: ABO/Functions/MPCs/TransformToDenseFormulation.m:30-91 (literal loops and products).
 * A[N][2][2], B[N][2][nu], D[N][2].  Outputs: Hd[nV][nV], cd[nV], Gd[nC][nV], lb/ub[nC],
 * Psi[nz][nV], d[nz], with nV = N*nu, nz = N*(2+nu)+2. */
void orc_transform_to_dense(int N, int nu, int nC, const double* A, const double* B,
                            const double* D, const double* Hs, const double* cs, const double* Gs,
                            const double* glb, const double* gub, double s_curr, double v_curr,
                            double* Hd, double* cd, double* Gd, double* lbd, double* ubd,
                            double* Psi, double* d);

/* A9: ABO/Functions/MPCs/RunPlantModel.m:27-44 */
void orc_run_plant_model(const eepacc_settings* S, const eepacc_vehicle* V, double s, double v,
                         double Fm, double Fb, double* s_next, double* v_next);
/* ABO/Functions/PWA_function_manipulation/InterpPWA.m:14-27 */
double orc_interp_pwa(double d, const double* doms, const double* vals, int n);
/* ABO/Functions/Other/GetMotorPower_FifthOrderSurface.m:16-20 */
double orc_motor_power_fifth(double Fm, double rpm, const double* b);

/* carried per-instance state of the closed loop */
typedef struct orc_loop_state {
    int    k;                 /* MPC step index kk (0-based)                               */
    double v_tv_measured;     /* previous lead speed (ABO/RunOpt_ABMPC.m:188)              */
    double s_prev_sol[EEPACC_MAX_HORIZON + 1], v_prev_sol[EEPACC_MAX_HORIZON + 1];
    double fbA22[EEPACC_MAX_HORIZON], fbD2[EEPACC_MAX_HORIZON];   /* FB A(k)/D(k) freeze      */
    double xwarm[8 * EEPACC_MAX_HORIZON];
} orc_loop_state;

typedef struct orc_step_io {
    /* in */
    double s, v, a_prev, t0, s_tv, v_tv, a_tv_prev;
    double v_prev, Fm_prev, Fb_prev;        /* FB only                                     */
    /* out */
    double out[EEPACC_OUT_N];
    double s_pred[EEPACC_MAX_HORIZON + 1], v_pred[EEPACC_MAX_HORIZON + 1];
    orc_qp_stats qp;
} orc_step_io;

/* One MPC step (A2..A8): ABO/RunOpt_ABMPC.m:193-329.  If dense_out != NULL the dense QP
 * (Hd, cd, Gd, lb, ub) of this step is copied there in that order (for H/G golden checks). */
int orc_ab_step(const eepacc_settings* S, const eepacc_vehicle* V, orc_step_io* io,
                double* dense_out);
/* ABO/RunOpt_FBMPC.m:204-320; st carries the A(k)/D(k) freeze quirk state (F3). */
int orc_fb_step(const eepacc_settings* S, const eepacc_vehicle* V, orc_loop_state* st,
                orc_step_io* io, double* dense_out);

/* Closed loops: ABO/RunOpt_ABMPC.m:154-340, ABO/RunOpt_FBMPC.m:161-331.  traj
 * [n_steps][EEPACC_OUT_N], status[n_steps]; s_tv,v_tv [n_steps]. */
int orc_run_abmpc(const eepacc_settings* S, const eepacc_vehicle* V, int n_steps, double s0,
                  double v0, double a_minus1, const double* s_tv, const double* v_tv,
                  double* traj, int* status, int* qp_iters);
int orc_run_fbmpc(const eepacc_settings* S, const eepacc_vehicle* V, int n_steps, double s0,
                  double v0, double a_minus1, const double* s_tv, const double* v_tv,
                  double* traj, int* status, int* qp_iters);

/* A10: ABO/RunOpt_ABMPC.m:343-349.  All arrays [n]. */
void orc_postprocess(const eepacc_settings* S, const eepacc_vehicle* V, int n, const double* v,
                     const double* Fm, double* rpm, double* Tm, double* P, double* E);

#ifdef __cplusplus
}
#endif
#endif
