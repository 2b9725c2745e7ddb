/*
 * eepacc.h -- C ABI of libeepacc, the MI355X-native batched EEPACC MPC engine.
 *
 * Drop-in boundary for the per-step RunOpt_ABMPC / RunOpt_FBMPC hot path of
 * stefavpolito/EEPACC_MPC_CasADi_MATLAB.  Every entry point cites the
 * reference interface it replaces.  Paths: ABO/ = ACCMPC-ABO_CasADi/, ORIG/ = MATLAB_CasADi/.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types.
 *   - all floating point data is IEEE fp64 (the reference is MATLAB double throughout).
 *   - batched arrays are batch-major structure-of-arrays: x[k*B + i] is item k (a horizon
 *     stage or a simulation step) of instance i, unit stride across instances.
 *   - "device" pointers are HIP device allocations on the handle's GPU (e.g. a torch
 *     tensor's data_ptr()); "host" entry points end in _host and copy in/out themselves.
 *   - every function returns 0 on success or a negative EEPACC_E* code.  A QP that does not
 *     converge is NOT an error: like the reference (opts.error_on_fail=false,
 *     ABO/RunOpt_ABMPC.m:121,255) the iterate is still applied and status[i] = 1 mirrors
 *     optSol.exitMessage(k).
 *   - a handle owns its device workspaces and per-instance warm-start state; it is
 *     thread-compatible (one handle per host thread / stream), not thread-safe, and admits ONE
 *     launch in flight at a time: the closed-loop kernels hand their loop state from work unit to
 *     work unit through buffers of the handle, so a second eepacc_run_* on another stream of the
 *     same handle must wait for the first (use one handle per concurrent stream).
 *   - status values: 0 solved, 1 QP not converged / infeasible (the reference's exitMessage),
 *     3 the step was not computed because a device-side wait timed out (then eepacc_synchronize
 *     returns EEPACC_EDEVICE; never expected in normal operation).
 */
#ifndef EEPACC_H
#define EEPACC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EEPACC_VERSION 1

#define EEPACC_OK            0
#define EEPACC_EINVAL       -1   /* malformed argument / unsupported setting            */
#define EEPACC_ENOMEM       -2   /* host or device allocation failed                   */
#define EEPACC_EDEVICE      -3   /* HIP runtime error (message: eepacc_last_error())   */
#define EEPACC_ENOTSUP      -4   /* setting valid in the reference but not built here  */

#define EEPACC_MAX_HORIZON   63  /* N_hor upper limit of the HIP kernels               */
#define EEPACC_QP_MAX_NV    384  /* dense QP operator: variables                       */
#define EEPACC_QP_MAX_NC   2048  /* dense QP operator: rows of A                       */

/* Vehicle constants: the struct V returned by SetVehicleParameters()
 * (ABO/Functions/Settings/SetVehicleParameters.m:12-133; ORIG/ holds the BMW i3 values). */
typedef struct eepacc_vehicle {
    double m, A_f, c_d, L, h_g, WD_s_F, L_f, L_r;
    double F0, F1, F2;             /* coast-down (ABO only)                             */
    double p00, p10, p01;          /* efficiency-map fit used by the fuel term (ABO)    */
    double P_m_max, T_m_max, omega_m_r, omega_m_max;
    double c_r, R_w, beta_gb, beta_fd, phi;
    double v_max;
    double eta_TF;
    double lambda, mu, rho_a, g, zeta_a;
    /* ICE fuel-map fit FC = k00 + k10*w_ICE + k01*T_ICE and the stepped gearbox it needs
     * (ABO/Functions/Settings/SetVehicleParameters.m:44-46,96-101; gear choice per horizon stage:
     * ABO/Functions/MPCs/LUTgearshift.m:17-41).  Read only when ab_fuel_term = 2. */
    double k00, k10, k01;
    double tau_fd, eta_drive;
    double upSpd[7];               /* upshift speed thresholds, m/s, ascending          */
    double tau_gb[8];              /* gearbox ratio of gears 1..8                       */
} eepacc_vehicle;

/* Controller / scenario settings: the fields of OPTsettings read on the hot path
 * (ABO/Settings.m, ABO/Functions/Settings/GenerateUseCase.m). */
typedef struct eepacc_settings {
    /* horizon (ABO/Settings.m:101-122, 243-250) */
    int32_t N_hor;
    const double*  Tvec;           /* [N_hor]  Tvec[0] is the controller sample time Ts */
    const int32_t* Mb;             /* [N_hor]  move-blocking mask, may be NULL (= zeros) */
    /* weights.  W_AB: ABO/Settings.m:48-64 has 7 entries [w_FC,w_a,w_j,w_v,w_h,w_s,w_f];
     * ORIG/Settings.m:48-62 has 6 (no w_FC): set ab_fuel_term = 0 and pass
     * [0,w_a,w_j,w_v,w_h,w_s,w_f].  W_FB: ABO/Settings.m:31-46 [w_P,w_a,w_j,w_v,w_h,w_s,w_f] */
    double  W_AB[7];
    double  W_FB[7];
    int32_t ab_fuel_term;          /* 1: efficiency-map fuel term ABO/.../CreateQP_AB.m:162-166;
                                    * 2: the ICE-map fuel term of :154-159 (what savedABMPCsolICEMAP.mat was
                                    *    written with): per-stage curvature 2 w_FC k01 F2 R_w/(tau_fd tau_est(k) eta_drive)
                                    *    with the gear ratio tau_est(k) = LUTgearshift(v_est(k)), so H changes every
                                    *    step; 0: no fuel term (ORIG)                                               */
    int32_t ab_route_rows;         /* 1: ORIG 18-row stage (speed/curve/stop/TL caps)    */
    /* vehicle following (ABO/Settings.m:203-204,143) */
    double  tau_min, h_min, s_goal;
    /* estimators (ABO/Settings.m:105-108) */
    int32_t paramEstSetting, TVestSetting;
    double  tConstACC_ego, tConstACC_tar;
    /* plant (ABO/Settings.m:111) */
    int32_t N_integratePlant;
    /* solver selection (ABO/Settings.m:98,114): 0 and 1 (sparse / dense qpOASES) pose the same QP and
     * are both accepted; 2 (HPIPM formulation with hard acceleration bounds) is not built */
    int32_t solverToUse;
    int32_t FBuseTaylor;
    /* power fits (ABO/Settings.m:229-238) */
    double  b_quadr[6];
    double  b_fifthOrder[21];
    /* route tables produced by GenerateUseCase (piecewise tables, knots ascending) */
    int32_t n_speedLim;  const double* s_speedLim;  const double* v_speedLim;
    int32_t n_curv;      const double* s_curv;      const double* curvature;
    int32_t n_slope;     const double* s_slope;     const double* slope;
    int32_t n_stop;      const double* stopLoc;
    int32_t n_TL;        const double* TLLoc;      /* [n_TL][4] row-major: loc,phase,red,green */
    double  stopRefDist, stopRefVelSlope, stopVel, TLstopVel, TLStopRegionSize, alpha_TTL;
    /* Baseline controller (ABO/RunOpt_BLMPC.m, ABO/Functions/MPCs/CreateQP_BL.m).  bl_mode = 1 makes a handle
     * a RunOpt_BLMPC: the ABMPC entry points (eepacc_ab_step, eepacc_run_abmpc, ...) then pose CreateQP_BL's problem
     * (n_u = 2: a, xi_f; one slack for all soft rows; objective -w_v sum v_k + w_a a^2 + w_j jerk^2 + w_f xi_f) with
     * the baseline comfort limits of EstimateRouteAndComfortBounds.m:173-189 (MPCtype 1).  The caller passes
     * N_hor = BL_N_hor, Tvec[k] = BL_Ts and paramEstSetting = BL_trajEstSett (ABO/Settings.m:137-139,
     * EstimateVehicleTrajectory.m:25-29); W_AB / W_FB are ignored.  W_BL = [w_v, w_a, w_j, w_f] (ABO/Settings.m:66-71).
     * With the reference's weights (w_a = w_j = 0) the problem is a linear program; the dual active set needs curvature,
     * so it is solved by the proximal-point iteration  a_{j+1} = argmin LP(a) + bl_lp_eps/2 |a - a_j|^2,  a_0 = 0
     * (bl_lp_eps <= 0: 0.1), each solve warm from the last working set, until the point stays (at most bl_prox_iter
     * re-centrings; 0: 40, < 0: none, i.e. the single regularised solve of earlier versions): for a linear program that
     * ends after finitely many steps at an optimum of the LP itself. */
    int32_t bl_mode, bl_prox_iter;
    double  W_BL[4];
    double  BL_a_LimLowVel, BL_a_LimHighVel, BL_j_LimLowVel, BL_j_LimHighVel;   /* ABO/Settings.m:131-134 */
    double  bl_lp_eps;
    /* ABMPC / baseline controller: a measured state that violates its own hard bounds (s_0 >= 0, 0 <= v_0 <= v_max,
     * CreateQP_AB.m:256-261, CreateQP_BL.m:214-222) by more than this makes the step infeasible (status 1).
     * <= 0: 1e-9 (baseline controller: 1e-5 m/s), which lets the closed loop's rounding noise at standstill through (the
     * saved ABMPC solutions have exitMessage = 0 there); 1e-11 reproduces the three bad exits of the saved baseline
     * solution (v_0 = -3e-10). */
    double  state_bound_tol;
} eepacc_settings;

typedef struct eepacc_handle eepacc_handle;

/* Number of doubles per instance in the per-step output block, batch-major [EEPACC_OUT_N][B]. */
enum {
    EEPACC_OUT_S = 0,     /* s_opt(k)   = z(1)  measured position the QP was solved at      */
    EEPACC_OUT_V,         /* v_opt(k)   = z(2)                                              */
    EEPACC_OUT_FM,        /* Fm_opt(k)  motor force after allocation (AB) / QP output (FB)  */
    EEPACC_OUT_FB,        /* Fb_opt(k)                                                      */
    EEPACC_OUT_A,         /* a_opt(k)   realised acceleration (ABO/RunOpt_ABMPC.m:324)      */
    EEPACC_OUT_XI_V, EEPACC_OUT_XI_H, EEPACC_OUT_XI_S, EEPACC_OUT_XI_F,
    EEPACC_OUT_COST,      /* sol.cost (dense-QP objective value, constant term excluded)    */
    EEPACC_OUT_DISTHOR,   /* DistHor(k) (ABO/RunOpt_ABMPC.m:200)                            */
    EEPACC_OUT_AQP,       /* QP stage-0 acceleration before allocation (z(3), AB only)      */
    EEPACC_OUT_N
};

const char* eepacc_last_error(void);
int  eepacc_version(void);
int  eepacc_sizeof_settings(void);   /* sizeof(eepacc_settings) as compiled: binding self-check */
int  eepacc_sizeof_vehicle(void);

/* Create / destroy.  Replaces the one-time set-up part of RunOpt_ABMPC / RunOpt_FBMPC
 * (ABO/RunOpt_ABMPC.m:14-123: unpack settings, state-space matrices, conic(...) creation).
 * device: HIP device ordinal.  max_batch: largest B used with this handle. */
int  eepacc_create(eepacc_handle** out, const eepacc_settings* S, const eepacc_vehicle* V,
                   int device, int max_batch);
void eepacc_destroy(eepacc_handle* h);

/* Reset the per-instance carried state (warm start, previous lead speed, FB A/D freeze,
 * step counter): the "kk == 0" branch of ABO/RunOpt_ABMPC.m:159-172. */
int  eepacc_reset(eepacc_handle* h);

/* B2 -- per-step operator: one receding-horizon step of ABMPC for B instances
 * (body of the kk-loop ABO/RunOpt_ABMPC.m:193-329 after the measurement block; the same
 * contract as the Simulink MATLAB-Function block ACCMPC(...) of ABO/ACCMPC.slx chart_121).
 * Inputs, device, each [B]: s, v (measured state), a_prev (a_minus1), t0, s_tv, v_tv,
 * a_tv_prev (lead acceleration estimate).  Outputs, device: out[EEPACC_OUT_N][B];
 * s_pred,v_pred [(N_hor+1)][B] = z(1:7:end), z(2:7:end) (may be NULL); status [B] int32.
 * stream: hipStream_t as void* (NULL = default stream).  Asynchronous on that stream. */
int  eepacc_ab_step(eepacc_handle* h, int B,
                    const double* s, const double* v, const double* a_prev, const double* t0,
                    const double* s_tv, const double* v_tv, const double* a_tv_prev,
                    double* out, double* s_pred, double* v_pred, int32_t* status,
                    void* stream);

/* B1 -- closed loop: optSol = RunOpt_ABMPC(OPTsettings) for B independent instances
 * (ABO/RunOpt_ABMPC.m:154-340 incl. measurement block, plant, force allocation).
 * n_steps = N_sim+1 iterations (kk = 0..N_sim).  Device inputs: s0,v0,a_minus1 [B];
 * s_tv,v_tv [n_steps][B] lead traces (already shifted by TVlength, ABO/Main.m:88).
 * Device outputs: traj [n_steps][EEPACC_OUT_N][B]; status [n_steps][B].
 * The handle carries the loop state: the first call after eepacc_create/eepacc_reset starts
 * at kk = 0, later calls continue where the previous one stopped (s_tv/v_tv then hold the
 * rows of the continued steps), so a long simulation can be run in chunks. */
int  eepacc_run_abmpc(eepacc_handle* h, int B, int n_steps,
                      const double* s0, const double* v0, const double* a_minus1,
                      const double* s_tv, const double* v_tv,
                      double* traj, int32_t* status, void* stream);

/* B3 -- dense QP operator: the call
 *   sol = QPsolver('h',H,'g',c,'a',G,'lbx',z_lb,'ubx',z_ub,'lba',g_lb,'uba',g_ub)
 * of ABO/RunOpt_ABMPC.m:252 and ABO/RunOpt_FBMPC.m:278 (CasADi conic, CAS/+casadi/conic.m:951-966)
 * for B independent problems of one shape:  min 1/2 x'Hx + g'x  s.t. lba <= Ax <= uba,
 * lbx <= x <= ubx.  Device arrays, instance-major: H [B][nV*nV] (symmetrised internally, so
 * row- or column-major), g [B][nV], A [B][nV][nC] = COLUMN-major nC x nV as MATLAB/CasADi hold
 * it, lba/uba [B][nC], lbx/ubx [B][nV]; +-inf entries and NULL bound arrays mean "absent".
 * x0 [B][nV] or NULL: proximal centre / initial guess (the reference passes none for AB and
 * hot-starts implicitly).  Outputs x [B][nV], cost [B] (may be NULL), status [B] (may be
 * NULL; 0 = KKT point verified, 1 = not converged).  H may be singular PSD or indefinite
 * (FB): a proximal term is added internally and removed by the final exact KKT solve. */
int  eepacc_qp_solve_batched(eepacc_handle* h, int B, int nV, int nC,
                             const double* H, const double* g, const double* A,
                             const double* lba, const double* uba,
                             const double* lbx, const double* ubx, const double* x0,
                             double* x, double* cost, int32_t* status, void* stream);

/* Same two operators for the force-based MPC (ABO/RunOpt_FBMPC.m:161-331).  v_prev, Fm_prev,
 * Fb_prev are the previous step's state/controls (ABO/RunOpt_FBMPC.m:188-191). */
int  eepacc_fb_step(eepacc_handle* h, int B,
                    const double* s, const double* v, const double* v_prev,
                    const double* a_prev, const double* Fm_prev, const double* Fb_prev,
                    const double* t0, const double* s_tv, const double* v_tv,
                    const double* a_tv_prev,
                    double* out, double* s_pred, double* v_pred, int32_t* status,
                    void* stream);
int  eepacc_run_fbmpc(eepacc_handle* h, int B, int n_steps,
                      const double* s0, const double* v0, const double* a_minus1,
                      const double* s_tv, const double* v_tv,
                      double* traj, int32_t* status, void* stream);

/* Host-pointer convenience wrappers (what a MEX gateway calls). */
int  eepacc_run_abmpc_host(eepacc_handle* h, int B, int n_steps,
                           const double* s0, const double* v0, const double* a_minus1,
                           const double* s_tv, const double* v_tv,
                           double* traj, int32_t* status);
int  eepacc_run_fbmpc_host(eepacc_handle* h, int B, int n_steps,
                           const double* s0, const double* v0, const double* a_minus1,
                           const double* s_tv, const double* v_tv,
                           double* traj, int32_t* status);

/* Baseline controller by name: optSol = RunOpt_BLMPC(OPTsettings) (ABO/RunOpt_BLMPC.m:1, ABO/Main.m:97) and the body of
 * its loop (:175-300).  Same arguments, outputs and error behaviour as eepacc_ab_step / eepacc_run_abmpc /
 * eepacc_run_abmpc_host; they require a handle created with bl_mode = 1 (EEPACC_EINVAL otherwise), so a binding for
 * RunOpt_BLMPC cannot silently run the ABMPC problem.  The xi_v, xi_h, xi_s entries of the output block are zero
 * (the baseline QP has the one slack xi_f). */
int  eepacc_bl_step(eepacc_handle* h, int B,
                    const double* s, const double* v, const double* a_prev, const double* t0,
                    const double* s_tv, const double* v_tv, const double* a_tv_prev,
                    double* out, double* s_pred, double* v_pred, int32_t* status, void* stream);
int  eepacc_run_blmpc(eepacc_handle* h, int B, int n_steps,
                      const double* s0, const double* v0, const double* a_minus1,
                      const double* s_tv, const double* v_tv,
                      double* traj, int32_t* status, void* stream);
int  eepacc_run_blmpc_host(eepacc_handle* h, int B, int n_steps,
                           const double* s0, const double* v0, const double* a_minus1,
                           const double* s_tv, const double* v_tv,
                           double* traj, int32_t* status);

/* Post-processing of a closed-loop trajectory (ABO/RunOpt_ABMPC.m:343-349): rpm, Tm,
 * fifth-order battery power P and cumulative energy E, all device [n_steps][B]. */
int  eepacc_postprocess(eepacc_handle* h, int B, int n_steps, const double* traj,
                        double* rpm, double* Tm, double* P, double* E, void* stream);

/* Solver statistics of the last launch, device [B]: active-set iterations used. */
int  eepacc_last_iterations(eepacc_handle* h, int B, int32_t* iters_host);

/* Wait for the work queued on `stream` and report the handle's sticky device error word:
 * EEPACC_EDEVICE if a closed-loop launch since the last eepacc_reset could not hand its loop state
 * on (affected steps carry status 3), EEPACC_OK otherwise.  The *_host wrappers call it themselves. */
int  eepacc_synchronize(eepacc_handle* h, void* stream);

/* Flag string the library was compiled with ("" for a release build; instrumented builds such as
 * -DEEPACC_AB_TIMING change the meaning of the iteration / status outputs). */
const char* eepacc_build_flags(void);

#ifdef __cplusplus
}
#endif
#endif /* EEPACC_H */
