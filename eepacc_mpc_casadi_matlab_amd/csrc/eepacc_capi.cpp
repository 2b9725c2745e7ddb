// eepacc_capi.cpp -- C-ABI of libeepacc (include/eepacc.h): handle management, validation of the
// reference settings, one-time host precomputation, kernel launches.  Compiled with hipcc.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "eepacc_device.h"
#include "eepacc_qp_dense.h"
#include "eepacc_fb.h"
#include "eepacc_fbs.h"
#include "../../include/eepacc.h"

namespace eepacc {
size_t ab_smem_bytes(int N);
size_t ab_hb_doubles(int N, int B, int num_cus);
hipError_t launch_ab_step(const DevCfg* dC, int N, int variant, int B, const double* s, const double* v, const double* a_prev,
                          const double* t0, const double* s_tv, const double* v_tv, const double* a_tv_prev,
                          unsigned long long* codes, double* out, double* s_pred, double* v_pred,
                          int32_t* status, int32_t* iters, hipStream_t stream);
hipError_t launch_run_abmpc(const DevCfg* dC, int N, int variant, int B, int k_start, int n_steps, const double* s0,
                            const double* v0, const double* a_m1, const double* s_tv, const double* v_tv,
                            double* carry, unsigned long long* codes, double* traj,
                            int32_t* status, int32_t* iters_total, int* work_counter, int* done, int* err_word, int num_cus,
                            hipStream_t stream);
hipError_t launch_postprocess(const DevCfg* dC, int B, int n_steps, const double* traj, double* rpm, double* Tm,
                              double* P, double* E, hipStream_t stream);
hipError_t set_max_smem();
int pick_chunk_steps(int n_steps, int B, int resident_waves);
}  // namespace eepacc

using eepacc::DevCfg;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
namespace eepacc { int set_error(int code, const std::string& msg) { return fail(code, msg); } }   // other translation units (eepacc_nlp.hip)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(EEPACC_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct eepacc_handle {
    int device = 0;
    int max_batch = 0;
    DevCfg cfg;
    DevCfg* d_cfg = nullptr;
    double* d_Hinv = nullptr;
    double* d_hb = nullptr;          // ICE variant: per-wave base inverse of the step
    double* d_pred = nullptr;                // [max_batch][2][64] previous predictions (paramEstSetting 2)
    unsigned long long* d_codes = nullptr;   // [max_batch][64]
    int32_t* d_iters = nullptr;              // [max_batch]
    double* d_carry = nullptr;               // [6][B] closed-loop carry (see k_run_abmpc)
    int* d_counter = nullptr;                // work counter of the closed-loop kernel
    int* d_done = nullptr;                   // [max_batch] chunks finished per instance
    int* d_err = nullptr;                    // sticky device error word (bit 0: a closed-loop hand-off timed out)
    int num_cus = 256;
    int last_B = 0;
    int k_done = 0;                          // closed-loop steps already run since the last reset
    int carry_B = 0;
    double* d_qp_ws = nullptr;               // workspace of the dense QP operator
    size_t qp_ws_doubles = 0;
    int* d_qp_counter = nullptr;
    // FBMPC state (allocated on first use)
    int fb_B = 0, fb_chunk = 0;
    int fb_k_done = 0;
    bool fb_by_step = false;                 // the FB step counter was advanced by eepacc_fb_step (no closed-loop carry to resume from)
    double *fb_H = nullptr, *fb_g = nullptr, *fb_A = nullptr, *fb_lba = nullptr, *fb_uba = nullptr;   // [fb_chunk]
    double *fb_x = nullptr, *fb_x0 = nullptr, *fb_cost = nullptr, *fb_meas = nullptr, *fb_carry = nullptr;   // [fb_B]
    double *fb_A22 = nullptr, *fb_D2 = nullptr;
    double *fb_sp = nullptr, *fb_vp = nullptr;   // [N+1][fb_B] predictions of the last step
    int32_t* fb_qpstat = nullptr;
    int* fb_rhok = nullptr;                  // [fb_B] regularisation exponent found at the previous step
    // structured FBMPC path (eepacc_fbs.hip): per-instance state, closed-loop carry, base-inverse scratch
    bool fbs = false;                        // settings are covered by the structured solver
    double *fbs_state = nullptr, *fbs_carry = nullptr, *fbs_hb = nullptr;
};

extern "C" const char* eepacc_last_error(void) { return g_err.c_str(); }
extern "C" int eepacc_version(void) { return EEPACC_VERSION; }
extern "C" int eepacc_sizeof_settings(void) { return (int)sizeof(eepacc_settings); }
extern "C" int eepacc_sizeof_vehicle(void) { return (int)sizeof(eepacc_vehicle); }

// symmetric positive definite inverse (Gauss-Jordan in long double; N <= 63)
static bool spd_inverse(std::vector<long double>& A, int n) {
    std::vector<long double> I((size_t)n * n, 0.0L);
    for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.0L;
    for (int k = 0; k < n; ++k) {
        long double d = A[(size_t)k * n + k];
        if (!(d > 0.0L)) return false;
        for (int j = 0; j < n; ++j) { A[(size_t)k * n + j] /= d; I[(size_t)k * n + j] /= d; }
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            long double f = A[(size_t)i * n + k];
            if (f == 0.0L) continue;
            for (int j = 0; j < n; ++j) { A[(size_t)i * n + j] -= f * A[(size_t)k * n + j]; I[(size_t)i * n + j] -= f * I[(size_t)k * n + j]; }
        }
    }
    A.swap(I);
    return true;
}

// kernel variant of a handle: 0 plain, 1 blocked moves, 2 baseline controller
static int ab_variant(const DevCfg& C) { return C.bl_mode ? 2 : (C.ab_fuel_term == 2 ? 3 : (C.mb_any ? 1 : 0)); }

static int build_cfg(const eepacc_settings* S, const eepacc_vehicle* V, DevCfg& C, std::vector<double>& Hinv) {
    memset(&C, 0, sizeof(C));
    const int N = S->N_hor;
    if (N < 2 || N > eepacc::kMaxN) return fail(EEPACC_EINVAL, "N_hor must be in [2, 63]");
    if (!S->Tvec) return fail(EEPACC_EINVAL, "Tvec is NULL");
    // solverToUse 0 (sparse qpOASES) states the same QP as 1 (dense qpOASES) with the dynamics kept as
    // equality rows (CreateQP_AB.m:227-246): same feasible set and objective, hence the same minimiser and
    // the same kernels.  2 (HPIPM) is a different problem (hard bounds a in [-8, 8], :79-99).
    if (S->solverToUse != 0 && S->solverToUse != 1)
        return fail(EEPACC_ENOTSUP, "solverToUse == 2 (HPIPM formulation, ABO/Settings.m:114) is not built");
    if (S->paramEstSetting < 0 || S->paramEstSetting > 2) return fail(EEPACC_EINVAL, "paramEstSetting must be 0, 1 or 2");
    if (S->TVestSetting != 0 && S->TVestSetting != 1) return fail(EEPACC_EINVAL, "TVestSetting must be 0 or 1");
    if (S->n_speedLim < 1 || S->n_speedLim > eepacc::kMaxKnots || S->n_curv < 1 || S->n_curv > eepacc::kMaxKnots ||
        S->n_slope < 1 || S->n_slope > eepacc::kMaxKnots || S->n_stop < 0 || S->n_stop > eepacc::kMaxStops ||
        S->n_TL < 0 || S->n_TL > eepacc::kMaxTL)
        return fail(EEPACC_EINVAL, "route table sizes out of range");
    if (S->N_integratePlant < 1) return fail(EEPACC_EINVAL, "N_integratePlant < 1");
    if (S->ab_fuel_term < 0 || S->ab_fuel_term > 2) return fail(EEPACC_EINVAL, "ab_fuel_term must be 0, 1 or 2");
    if (S->ab_fuel_term == 2 && !S->bl_mode) {
        // ICE-map fuel term (CreateQP_AB.m:154-159): step-varying Hessian, built and inverted in LDS by its own kernel variant
        if (N > 32) return fail(EEPACC_ENOTSUP, "ab_fuel_term == 2 (ICE-map fuel term) is built for N_hor <= 32");
        for (int k = 0; S->Mb && k < N; ++k)
            if (S->Mb[k] != 0) return fail(EEPACC_ENOTSUP, "ab_fuel_term == 2 (ICE-map fuel term) is not built with move blocking");
        if (!(V->tau_fd > 0.0) || !(V->eta_drive > 0.0) || !(V->R_w > 0.0))
            return fail(EEPACC_EINVAL, "ab_fuel_term == 2 needs V.tau_fd, V.eta_drive, V.R_w > 0 (SetVehicleParameters.m:92,100-101)");
        for (int g2 = 0; g2 < 8; ++g2) if (!(V->tau_gb[g2] > 0.0)) return fail(EEPACC_EINVAL, "ab_fuel_term == 2 needs positive gear ratios V.tau_gb");
        for (int g2 = 1; g2 < 7; ++g2) if (!(V->upSpd[g2] >= V->upSpd[g2 - 1])) return fail(EEPACC_EINVAL, "V.upSpd must ascend");
    }
    C.N = N;
    C.ab_fuel_term = S->ab_fuel_term; C.ab_route_rows = S->ab_route_rows;
    C.paramEstSetting = S->paramEstSetting; C.TVestSetting = S->TVestSetting;
    C.N_integratePlant = S->N_integratePlant;
    C.max_iter = 60 * N + 200;
    C.const_T = 1;
    C.tau[0] = 0.0;
    for (int k = 0; k < N; ++k) {
        if (!(S->Tvec[k] > 0.0)) return fail(EEPACC_EINVAL, "Tvec entries must be positive");
        C.Tvec[k] = S->Tvec[k];
        C.tau[k + 1] = C.tau[k] + S->Tvec[k];
        if (S->Tvec[k] != S->Tvec[0]) C.const_T = 0;
        if (S->Mb && S->Mb[k] != 0) C.mb_any = 1;
    }
    if (C.mb_any) {
        // block structure: stage k with Mb[k] = 1 repeats the acceleration of the previous stage
        if (S->Mb[0] != 0) return fail(EEPACC_EINVAL, "Mb[0] must be 0 (the first stage has no predecessor in the horizon)");
        int maxlen = 1;
        for (int k = 0, lead = 0; k < N; ++k) {
            if (S->Mb[k] != 0 && S->Mb[k] != 1) return fail(EEPACC_EINVAL, "Mb entries must be 0 or 1");
            if (S->Mb[k] == 0) lead = k;
            C.mb_lead[k] = lead;
            C.mb_end[lead] = k;
            if (k - lead + 1 > maxlen) maxlen = k - lead + 1;
        }
        for (int k = 0; k < N; ++k) if (C.mb_lead[k] != k) C.mb_end[k] = k;
        C.mb_lead[N] = N; C.mb_end[N] = N;
        C.mb_maxlen = maxlen;
    } else {
        for (int k = 0; k <= N; ++k) { C.mb_lead[k] = k; C.mb_end[k] = k; }
        C.mb_maxlen = 1;
    }
    C.fb_row0[0] = 0;
    for (int k = 0; k < N; ++k) {
        C.mb_mask[k] = (S->Mb && S->Mb[k] == 1) ? 1 : 0;
        C.fb_row0[k + 1] = C.fb_row0[k] + 26 + 2 * C.mb_mask[k];
    }
    C.mb_mask[N] = 0;
    C.w_FC = S->ab_fuel_term ? S->W_AB[0] : 0.0;
    C.w_a = S->W_AB[1]; C.w_j = S->W_AB[2]; C.w_v = S->W_AB[3]; C.w_h = S->W_AB[4]; C.w_s = S->W_AB[5]; C.w_f = S->W_AB[6];
    // default: above the noise level of the closed loop at standstill -- ABMPC 1e-10; the baseline LP (solved with the
    // curvature 1e-4) leaves 2e-9 on plain stops and up to 1.5e-6 when its slack is in play (use case 12)
    C.state_tol = S->state_bound_tol > 0.0 ? S->state_bound_tol : (S->bl_mode ? 1e-5 : 1e-9);
    double bl_travel = 0.0;
    if (S->bl_mode) {
        // the LP needs about 13 working-set changes per warm step and 3N from cold (871 saved steps at N = 20 as cold QPs:
        // mean 58, 99th percentile 113, maximum 270); a solve that is still going after 15N + 60 is cycling on a degenerate
        // vertex and is cut off there
        C.max_iter = 15 * N + 60;
        if (const char* ev = getenv("EEPACC_DEBUG_BL_MAX_ITER")) C.max_iter = atoi(ev);
        // RunOpt_BLMPC: CreateQP_BL.m:36-39,131-148  W_BL = [w_v (travel incentive), w_a, w_j, w_f]
        if (S->bl_mode != 1) return fail(EEPACC_EINVAL, "bl_mode must be 0 or 1");
        if (C.mb_any) return fail(EEPACC_ENOTSUP, "the baseline controller has no move blocking (RunOpt_BLMPC.m)");
        if (S->W_BL[0] < 0 || S->W_BL[1] < 0 || S->W_BL[2] < 0 || !(S->W_BL[3] > 0))
            return fail(EEPACC_EINVAL, "W_BL weights must be non-negative (w_f positive)");
        C.bl_mode = 1;
        C.ab_fuel_term = 0; C.ab_route_rows = 1;            // CreateQP_BL.m:264-288: the four speed caps are always present
        C.w_FC = 0.0; C.w_a = S->W_BL[1]; C.w_j = S->W_BL[2]; C.w_f = S->W_BL[3];
        C.w_v = 0.0; C.w_s = 0.0; C.w_h = 1.0;              // groups that do not exist in the baseline QP
        bl_travel = S->W_BL[0];
        C.bl_eps = (C.w_a == 0.0 && C.w_j == 0.0) ? (S->bl_lp_eps > 0.0 ? S->bl_lp_eps : 0.1) : 0.0;
        if (const char* ev = getenv("EEPACC_DEBUG_BL_EPS")) { if (C.bl_eps > 0.0 && atof(ev) > 0.0) C.bl_eps = atof(ev); }
        C.bl_prox_max = C.bl_eps > 0.0 ? (S->bl_prox_iter == 0 ? 40 : (S->bl_prox_iter < 0 ? 0 : S->bl_prox_iter)) : 0;
        C.bl_aLo = S->BL_a_LimLowVel; C.bl_aHi = S->BL_a_LimHighVel; C.bl_jLo = S->BL_j_LimLowVel; C.bl_jHi = S->BL_j_LimHighVel;
        if (!(C.bl_aLo > 0) || !(C.bl_aHi > 0) || !(C.bl_jLo > 0) || !(C.bl_jHi > 0))
            return fail(EEPACC_EINVAL, "baseline acceleration / jerk limits must be positive");
    } else if (!(C.w_a > 0.0) || !(C.w_h > 0.0) || C.w_v < 0 || C.w_s < 0 || C.w_f < 0 || C.w_j < 0)
        return fail(EEPACC_EINVAL, "W_AB weights must be positive (w_a, w_h) / non-negative");
    C.tau_min = S->tau_min; C.h_min = S->h_min; C.s_goal = S->s_goal;
    C.tConstACC_ego = S->tConstACC_ego; C.tConstACC_tar = S->tConstACC_tar;
    C.m = V->m; C.lambda = V->lambda; C.g = V->g; C.zeta_a = V->zeta_a; C.c_r = V->c_r; C.mu = V->mu; C.L = V->L;
    C.L_f = V->L_f; C.h_g = V->h_g; C.phi = V->phi; C.T_m_max = V->T_m_max; C.P_m_max = V->P_m_max;
    C.eta_TF = V->eta_TF; C.omega_m_r = V->omega_m_r; C.v_max = V->v_max;
    C.p01 = V->p01; C.p10 = V->p10; C.F2 = V->F2;
    C.cq = C.w_FC * V->p01 * V->F2;
    C.glin_v = C.bl_mode ? -bl_travel : C.w_FC * V->p10;
    C.glin_a = C.w_FC * V->p01 * V->lambda * V->m;
    if (C.ab_fuel_term == 2) {
        // per stage: cq_k = ice_cq / tau_k, lv_k = ice_lv tau_k, la_k = ice_la / tau_k (CreateQP_AB.m:154-159)
        C.cq = 0.0; C.glin_v = 0.0; C.glin_a = 0.0;
        C.ice_cq = C.w_FC * V->k01 * V->F2 * V->R_w / V->tau_fd / V->eta_drive;
        C.ice_lv = C.w_FC * V->k10 / V->R_w * V->tau_fd;
        C.ice_la = C.w_FC * V->k01 * V->lambda * V->m * V->R_w / V->tau_fd / V->eta_drive;
        for (int g2 = 0; g2 < 7; ++g2) C.ice_up[g2] = V->upSpd[g2];
        for (int g2 = 0; g2 < 8; ++g2) C.ice_gb[g2] = V->tau_gb[g2];
    }
    C.n_speedLim = S->n_speedLim; C.n_curv = S->n_curv; C.n_slope = S->n_slope; C.n_stop = S->n_stop; C.n_TL = S->n_TL;
    for (int i = 0; i < S->n_speedLim; ++i) { C.s_speedLim[i] = S->s_speedLim[i]; C.v_speedLim[i] = S->v_speedLim[i]; }
    const double alpha = S->alpha_TTL;
    for (int i = 0; i < S->n_curv; ++i) {
        C.s_curv[i] = S->s_curv[i];
        C.vcurv_tab[i] = alpha * pow(fabs(S->curvature[i]), -1.0 / 3.0);   // EstimateRouteAndComfortBounds.m:106,108
    }
    C.const_slope = 1;
    for (int i = 0; i < S->n_slope; ++i) {
        C.s_slope[i] = S->s_slope[i]; C.slope[i] = S->slope[i];
        if (S->slope[i] != S->slope[0]) C.const_slope = 0;
    }
    C.theta0 = S->slope[0]; C.sin_theta0 = sin(C.theta0); C.cos_theta0 = cos(C.theta0);
    for (int i = 0; i < S->n_stop; ++i) C.stopLoc[i] = S->stopLoc[i];
    for (int i = 0; i < 4 * S->n_TL; ++i) C.TLLoc[i] = S->TLLoc[i];
    C.stopRefDist = S->stopRefDist; C.stopRefVelSlope = S->stopRefVelSlope; C.stopVel = S->stopVel;
    C.TLstopVel = S->TLstopVel; C.TLStopRegionSize = S->TLStopRegionSize;
    for (int i = 0; i < 21; ++i) C.b5[i] = S->b_fifthOrder[i];
    for (int i = 0; i < 7; ++i) C.fb_w[i] = S->W_FB[i];
    for (int i = 0; i < 6; ++i) C.b_quadr[i] = S->b_quadr[i];
    C.FBuseTaylor = S->FBuseTaylor ? 1 : 0;
    // a-space Hessian of the condensed objective (step invariant for AB):
    //   H = 2 cq Sv'Sv + 2 w_a I + jerk tridiagonal (CreateQP_AB.m:162-180 through Psi)
    std::vector<long double> H((size_t)N * N, 0.0L);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            // sum over stages k = max(i,j)+1 .. N-1 of T_i T_j
            int cnt = N - 1 - (i > j ? i : j);
            if (cnt > 0) H[(size_t)i * N + j] += 2.0L * C.cq * C.Tvec[i] * C.Tvec[j] * cnt;
        }
    for (int k = 0; k < N; ++k) {
        H[(size_t)k * N + k] += 2.0L * C.w_a + (long double)C.bl_eps;
        long double qj = 2.0L * C.w_j / ((long double)C.Tvec[k] * C.Tvec[k]);
        H[(size_t)k * N + k] += qj;
        if (k > 0) {
            H[(size_t)(k - 1) * N + (k - 1)] += qj;
            H[(size_t)k * N + (k - 1)] -= qj;
            H[(size_t)(k - 1) * N + k] -= qj;
        }
    }
    if (C.mb_any) {
        // reduced variables (one acceleration per block): Hbar = E'HE on the leaders, identity on the rest
        std::vector<long double> Hb((size_t)N * N, 0.0L);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j)
                Hb[(size_t)C.mb_lead[i] * N + C.mb_lead[j]] += H[(size_t)i * N + j];
        for (int k = 0; k < N; ++k) if (C.mb_lead[k] != k) Hb[(size_t)k * N + k] = 1.0L;
        H.swap(Hb);
    }
    if (!spd_inverse(H, N)) return fail(EEPACC_EINVAL, "condensed Hessian is not positive definite");
    Hinv.assign((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            Hinv[(size_t)i * N + j] = (double)(0.5L * (H[(size_t)i * N + j] + H[(size_t)j * N + i]));
    if (eepacc::ab_smem_bytes(N) > 157 * 1024)
        return fail(EEPACC_ENOTSUP, "N_hor too large for the LDS layout of this build");
    return EEPACC_OK;
}

extern "C" int eepacc_create(eepacc_handle** out, const eepacc_settings* S, const eepacc_vehicle* V,
                             int device, int max_batch) {
    if (!out || !S || !V || max_batch < 1) return fail(EEPACC_EINVAL, "eepacc_create: bad arguments");
    *out = nullptr;
    DevCfg C;
    std::vector<double> Hinv;
    int rc = build_cfg(S, V, C, Hinv);
    if (rc != EEPACC_OK) return rc;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(EEPACC_EDEVICE, "no HIP device: libeepacc has no CPU path");
    if (device < 0 || device >= ndev) return fail(EEPACC_EINVAL, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    // the handle is destroyed (and everything allocated so far freed) if any later step fails
    struct Guard { eepacc_handle* h; ~Guard() { if (h) eepacc_destroy(h); } } guard{new eepacc_handle()};
    eepacc_handle* h = guard.h;
    h->device = device; h->max_batch = max_batch;
    HIPCHK(hipMalloc(&h->d_Hinv, Hinv.size() * sizeof(double)));
    HIPCHK(hipMemcpy(h->d_Hinv, Hinv.data(), Hinv.size() * sizeof(double), hipMemcpyHostToDevice));
    C.Hinv = h->d_Hinv;
    HIPCHK(hipMalloc(&h->d_pred, (size_t)max_batch * 128 * sizeof(double)));
    HIPCHK(hipMemset(h->d_pred, 0, (size_t)max_batch * 128 * sizeof(double)));
    C.pred = h->d_pred;
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (C.ab_fuel_term == 2 && !C.bl_mode) {
        HIPCHK(hipMalloc(&h->d_hb, eepacc::ab_hb_doubles(C.N, max_batch, h->num_cus) * sizeof(double)));
        C.hb = h->d_hb;
    }
    h->cfg = C;
    HIPCHK(hipMalloc(&h->d_cfg, sizeof(DevCfg)));
    HIPCHK(hipMemcpy(h->d_cfg, &C, sizeof(DevCfg), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->d_codes, (size_t)max_batch * 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(h->d_codes, 0, (size_t)max_batch * 64 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&h->d_iters, (size_t)max_batch * sizeof(int32_t)));
    HIPCHK(hipMemset(h->d_iters, 0, (size_t)max_batch * sizeof(int32_t)));
    HIPCHK(hipMalloc(&h->d_carry, (size_t)max_batch * 6 * sizeof(double)));
    HIPCHK(hipMemset(h->d_carry, 0, (size_t)max_batch * 6 * sizeof(double)));
    HIPCHK(hipMalloc(&h->d_counter, sizeof(int)));
    HIPCHK(hipMalloc(&h->d_done, sizeof(int) * (size_t)max_batch));
    HIPCHK(hipMalloc(&h->d_err, sizeof(int)));
    HIPCHK(hipMemset(h->d_err, 0, sizeof(int)));
    HIPCHK(hipMalloc(&h->d_qp_counter, sizeof(int)));
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    HIPCHK(eepacc::set_max_smem());
    // FBMPC: structured kernels unless the settings need the dense path (or EEPACC_FB_DENSE=1 asks for it)
    h->fbs = eepacc::fbs_supported(C) && eepacc::fbs_smem_bytes(C.N) <= 160 * 1024 - 4608;
    if (const char* e = getenv("EEPACC_FB_DENSE")) if (atoi(e) != 0) h->fbs = false;
    if (h->fbs) {
        HIPCHK(eepacc::fbs_set_max_smem());
        const size_t ns = (size_t)max_batch * eepacc::kFbsStateDoubles;
        HIPCHK(hipMalloc(&h->fbs_state, ns * sizeof(double)));
        HIPCHK(hipMemset(h->fbs_state, 0, ns * sizeof(double)));
        HIPCHK(hipMalloc(&h->fbs_carry, (size_t)max_batch * 6 * sizeof(double)));
        HIPCHK(hipMemset(h->fbs_carry, 0, (size_t)max_batch * 6 * sizeof(double)));
        HIPCHK(hipMalloc(&h->fbs_hb, eepacc::fbs_hb_doubles(C.N, max_batch, h->num_cus) * sizeof(double)));
    }
    guard.h = nullptr;
    *out = h;
    return EEPACC_OK;
}

static void fb_free(eepacc_handle* h) {
    double** ptrs[] = {&h->fb_H, &h->fb_g, &h->fb_A, &h->fb_lba, &h->fb_uba, &h->fb_x, &h->fb_x0, &h->fb_cost,
                       &h->fb_meas, &h->fb_carry, &h->fb_A22, &h->fb_D2, &h->fb_sp, &h->fb_vp};
    for (double** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (h->fb_qpstat) (void)hipFree(h->fb_qpstat);
    h->fb_qpstat = nullptr;
    if (h->fb_rhok) (void)hipFree(h->fb_rhok);
    h->fb_rhok = nullptr;
    h->fb_B = 0; h->fb_chunk = 0;
}

extern "C" void eepacc_destroy(eepacc_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_cfg) (void)hipFree(h->d_cfg);
    if (h->d_Hinv) (void)hipFree(h->d_Hinv);
    if (h->d_hb) (void)hipFree(h->d_hb);
    if (h->d_pred) (void)hipFree(h->d_pred);
    if (h->d_codes) (void)hipFree(h->d_codes);
    if (h->d_iters) (void)hipFree(h->d_iters);
    if (h->d_carry) (void)hipFree(h->d_carry);
    if (h->d_counter) (void)hipFree(h->d_counter);
    if (h->d_done) (void)hipFree(h->d_done);
    if (h->d_err) (void)hipFree(h->d_err);
    if (h->fbs_state) (void)hipFree(h->fbs_state);
    if (h->fbs_carry) (void)hipFree(h->fbs_carry);
    if (h->fbs_hb) (void)hipFree(h->fbs_hb);
    if (h->d_qp_ws) (void)hipFree(h->d_qp_ws);
    if (h->d_qp_counter) (void)hipFree(h->d_qp_counter);
    fb_free(h);
    delete h;
}

extern "C" int eepacc_reset(eepacc_handle* h) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemset(h->d_codes, 0, (size_t)h->max_batch * 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(h->d_pred, 0, (size_t)h->max_batch * 128 * sizeof(double)));
    HIPCHK(hipMemset(h->d_err, 0, sizeof(int)));
    h->k_done = 0; h->carry_B = 0;
    h->fb_k_done = 0; h->fb_by_step = false;
    if (h->fbs_state) HIPCHK(hipMemset(h->fbs_state, 0, (size_t)h->max_batch * eepacc::kFbsStateDoubles * sizeof(double)));
    if (h->fb_x0) HIPCHK(hipMemset(h->fb_x0, 0, (size_t)h->fb_B * 6 * h->cfg.N * sizeof(double)));
    if (h->fb_sp) {
        HIPCHK(hipMemset(h->fb_sp, 0, (size_t)h->fb_B * (h->cfg.N + 1) * sizeof(double)));
        HIPCHK(hipMemset(h->fb_vp, 0, (size_t)h->fb_B * (h->cfg.N + 1) * sizeof(double)));
    }
    return EEPACC_OK;
}

extern "C" int eepacc_ab_step(eepacc_handle* h, int B, const double* s, const double* v, const double* a_prev,
                              const double* t0, const double* s_tv, const double* v_tv, const double* a_tv_prev,
                              double* out, double* s_pred, double* v_pred, int32_t* status, void* stream) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    if (B < 0 || B > h->max_batch) return fail(EEPACC_EINVAL, "B exceeds max_batch of the handle");
    if (B == 0) return EEPACC_OK;
    if (!s || !v || !a_prev || !t0 || !s_tv || !v_tv || !a_tv_prev || !out || !status)
        return fail(EEPACC_EINVAL, "eepacc_ab_step: NULL buffer");
    HIPCHK(hipSetDevice(h->device));
    h->last_B = B;
    HIPCHK(eepacc::launch_ab_step(h->d_cfg, h->cfg.N, ab_variant(h->cfg), B, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, h->d_codes, out,
                                  s_pred, v_pred, status, h->d_iters, (hipStream_t)stream));
    return EEPACC_OK;
}

extern "C" int eepacc_run_abmpc(eepacc_handle* h, int B, int n_steps, const double* s0, const double* v0,
                                const double* a_minus1, const double* s_tv, const double* v_tv, double* traj,
                                int32_t* status, void* stream) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    if (B < 0 || B > h->max_batch || n_steps < 0) return fail(EEPACC_EINVAL, "bad B / n_steps");
    if (B == 0 || n_steps == 0) return EEPACC_OK;
    if (!s0 || !v0 || !a_minus1 || !s_tv || !v_tv || !traj || !status)
        return fail(EEPACC_EINVAL, "eepacc_run_abmpc: NULL buffer");
    HIPCHK(hipSetDevice(h->device));
    if (h->k_done > 0 && h->carry_B != B)
        return fail(EEPACC_EINVAL, "eepacc_run_abmpc: B changed while resuming; call eepacc_reset first");
    h->last_B = B;
    HIPCHK(eepacc::launch_run_abmpc(h->d_cfg, h->cfg.N, ab_variant(h->cfg), B, h->k_done, n_steps, s0, v0, a_minus1, s_tv, v_tv,
                                    h->d_carry, h->d_codes, traj, status, h->d_iters, h->d_counter, h->d_done, h->d_err, h->num_cus,
                                    (hipStream_t)stream));
    h->k_done += n_steps; h->carry_B = B;
    return EEPACC_OK;
}

extern "C" int eepacc_postprocess(eepacc_handle* h, int B, int n_steps, const double* traj, double* rpm,
                                  double* Tm, double* P, double* E, void* stream) {
    if (!h || !traj || !rpm || !Tm || !P || !E) return fail(EEPACC_EINVAL, "eepacc_postprocess: NULL argument");
    if (B < 1 || n_steps < 1) return EEPACC_OK;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(eepacc::launch_postprocess(h->d_cfg, B, n_steps, traj, rpm, Tm, P, E, (hipStream_t)stream));
    return EEPACC_OK;
}

extern "C" int eepacc_last_iterations(eepacc_handle* h, int B, int32_t* iters_host) {
    if (!h || !iters_host || B < 0 || B > h->max_batch) return fail(EEPACC_EINVAL, "eepacc_last_iterations: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(iters_host, h->d_iters, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost));
    return EEPACC_OK;
}

extern "C" int eepacc_synchronize(eepacc_handle* h, void* stream) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int err = 0;
    HIPCHK(hipMemcpy(&err, h->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (err != 0)
        return fail(EEPACC_EDEVICE, "closed-loop kernel: a work unit waited for its predecessor beyond the spin limit; "
                                    "the affected steps carry status 3 and the results of this launch are invalid");
    return EEPACC_OK;
}

#ifndef EEPACC_BUILD_FLAGS
#define EEPACC_BUILD_FLAGS ""
#endif
extern "C" const char* eepacc_build_flags(void) { return EEPACC_BUILD_FLAGS; }

// B3 -- dense QP operator (ABO/RunOpt_ABMPC.m:252)
// persistent workgroups of the dense QP kernel per compute unit (EEPACC_QP_WGS_PER_CU overrides)
static int qp_grid(const eepacc_handle* h, int B) {
    int per_cu = 2;
    if (const char* e = getenv("EEPACC_QP_WGS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) per_cu = v; }
    const int g = per_cu * h->num_cus;
    return B < g ? B : g;
}

static int qp_workspace(eepacc_handle* h, int grid, int nV) {
    size_t need = (size_t)grid * eepacc_qp_dense_ws_doubles(nV);
    if (need > h->qp_ws_doubles) {
        if (h->d_qp_ws) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(h->d_qp_ws); h->d_qp_ws = nullptr; h->qp_ws_doubles = 0; }
        if (hipMalloc(&h->d_qp_ws, need * sizeof(double)) != hipSuccess) return fail(EEPACC_ENOMEM, "dense QP workspace allocation failed");
        h->qp_ws_doubles = need;
    }
    return EEPACC_OK;
}

extern "C" int eepacc_qp_solve_batched(eepacc_handle* h, int B, int nV, int nC, const double* H, const double* g,
                                       const double* A, const double* lba, const double* uba, const double* lbx,
                                       const double* ubx, const double* x0, double* x, double* cost,
                                       int32_t* status, void* stream) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    if (B < 0 || nV < 1 || nC < 0) return fail(EEPACC_EINVAL, "eepacc_qp_solve_batched: bad sizes");
    if (B == 0) return EEPACC_OK;
    if (nV > EEPACC_QP_MAX_NV || nC > EEPACC_QP_MAX_NC)
        return fail(EEPACC_EINVAL, "eepacc_qp_solve_batched: nV/nC above EEPACC_QP_MAX_NV/NC");
    if (!H || !g || !x || (nC > 0 && !A)) return fail(EEPACC_EINVAL, "eepacc_qp_solve_batched: NULL buffer");
    if (eepacc_qp_dense_lds_bytes(nV, nC) > 160 * 1024) return fail(EEPACC_EINVAL, "eepacc_qp_solve_batched: problem does not fit LDS");
    HIPCHK(hipSetDevice(h->device));
    int grid = qp_grid(h, B);
    int rc = qp_workspace(h, grid, nV);
    if (rc != EEPACC_OK) return rc;
    eepacc_qp_args a;
    a.B = B; a.nV = nV; a.nC = nC; a.H = H; a.g = g; a.A = A; a.lba = lba; a.uba = uba; a.lbx = lbx; a.ubx = ubx;
    a.x0 = x0; a.x = x; a.cost = cost; a.status = status; a.iters = (B <= h->max_batch) ? h->d_iters : nullptr;
    a.ws = h->d_qp_ws; a.ws_stride = eepacc_qp_dense_ws_doubles(nV); a.rho_rel = 0.0; a.max_prox = 0;
    a.counter = h->d_qp_counter; a.rho_k = nullptr;
    HIPCHK(hipMemsetAsync(h->d_qp_counter, 0, sizeof(int), (hipStream_t)stream));
    HIPCHK(eepacc_qp_dense_launch(a, grid, (hipStream_t)stream));
    return EEPACC_OK;
}

// FBMPC (ABO/RunOpt_FBMPC.m:161-331): build kernel -> dense QP operator -> extraction, per step.
static int fb_prepare(eepacc_handle* h, int B) {
    const int N = h->cfg.N;
    const size_t nV = 6 * (size_t)N, nC = (size_t)h->cfg.fb_row0[N] + 2;
    if (nV > EEPACC_QP_MAX_NV || nC > EEPACC_QP_MAX_NC || eepacc_qp_dense_lds_bytes((int)nV, (int)nC) > 160 * 1024)
        return fail(EEPACC_ENOTSUP, "FBMPC: horizon too long for the dense QP operator");
    if (B <= h->fb_B) return EEPACC_OK;
    HIPCHK(hipDeviceSynchronize());
    fb_free(h);
    // the dense QP data is held for a chunk of instances at a time (about 16 GB at most)
    const size_t per = (nV * nV + nC * nV + nV + 2 * nC) * sizeof(double);
    size_t chunk = (size_t)16e9 / per;
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)B) chunk = (size_t)B;
    const size_t nB = (size_t)B;
    bool ok = hipMalloc(&h->fb_H, chunk * nV * nV * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_g, chunk * nV * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_A, chunk * nC * nV * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_lba, chunk * nC * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_uba, chunk * nC * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_x, nB * nV * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_x0, nB * nV * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_cost, nB * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_meas, 5 * nB * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_carry, 5 * nB * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_A22, nB * N * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_D2, nB * N * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_sp, nB * (N + 1) * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_vp, nB * (N + 1) * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->fb_qpstat, nB * sizeof(int32_t)) == hipSuccess &&
              hipMalloc(&h->fb_rhok, nB * sizeof(int)) == hipSuccess;
    if (!ok) { fb_free(h); return fail(EEPACC_ENOMEM, "FBMPC: device allocation failed"); }
    HIPCHK(hipMemset(h->fb_x0, 0, nB * nV * sizeof(double)));
    HIPCHK(hipMemset(h->fb_rhok, 0, nB * sizeof(int)));
    HIPCHK(hipMemset(h->fb_A22, 0, nB * N * sizeof(double)));
    HIPCHK(hipMemset(h->fb_D2, 0, nB * N * sizeof(double)));
    HIPCHK(hipMemset(h->fb_sp, 0, nB * (N + 1) * sizeof(double)));
    HIPCHK(hipMemset(h->fb_vp, 0, nB * (N + 1) * sizeof(double)));
    h->fb_B = B; h->fb_chunk = (int)chunk;
    h->fb_k_done = 0; h->fb_by_step = false;
    return EEPACC_OK;
}

// one FBMPC step for B instances; in[] as eepacc_fb_args documents for the mode
static int fb_one_step(eepacc_handle* h, int B, int mode, const double* s, const double* v, const double* a_prev,
                       const double* t0, const double* s_tv, const double* v_tv, const double* a_tv_prev,
                       double* out, double* s_pred, double* v_pred, int32_t* status, hipStream_t stream) {
    const int N = h->cfg.N, nV = 6 * N, nC = h->cfg.fb_row0[N] + 2;
    for (int b0 = 0; b0 < B; b0 += h->fb_chunk) {
        const int nb = (B - b0 < h->fb_chunk) ? B - b0 : h->fb_chunk;
        eepacc::eepacc_fb_args a;
        a.cfg = h->d_cfg; a.B = B; a.k_step = h->fb_k_done; a.b0 = b0; a.nb = nb; a.mode = mode;
        a.s = s; a.v = v; a.a_prev = a_prev; a.t0 = t0; a.s_tv = s_tv; a.v_tv = v_tv; a.a_tv_prev = a_tv_prev;
        a.carry = h->fb_carry; a.A22 = h->fb_A22; a.D2 = h->fb_D2;
        a.sp_prev = h->fb_sp; a.vp_prev = h->fb_vp;
        a.H = h->fb_H; a.g = h->fb_g; a.A = h->fb_A; a.lba = h->fb_lba; a.uba = h->fb_uba; a.meas = h->fb_meas;
        HIPCHK(eepacc::launch_fb_build(a, N, stream));
        int grid = qp_grid(h, nb);
        int rc = qp_workspace(h, grid, nV);
        if (rc != EEPACC_OK) return rc;
        eepacc_qp_args q;
        q.B = nb; q.nV = nV; q.nC = nC; q.H = h->fb_H; q.g = h->fb_g; q.A = h->fb_A; q.lba = h->fb_lba; q.uba = h->fb_uba;
        q.lbx = nullptr; q.ubx = nullptr;
        q.x0 = h->fb_x0 + (size_t)b0 * nV; q.x = h->fb_x + (size_t)b0 * nV; q.cost = h->fb_cost + b0;
        q.status = h->fb_qpstat + b0; q.iters = (B <= h->max_batch) ? h->d_iters + b0 : nullptr;
        q.ws = h->d_qp_ws; q.ws_stride = eepacc_qp_dense_ws_doubles(nV); q.rho_rel = 0.0; q.max_prox = 0;
        q.counter = h->d_qp_counter; q.rho_k = h->fb_rhok + b0;
        HIPCHK(hipMemsetAsync(h->d_qp_counter, 0, sizeof(int), stream));
        HIPCHK(eepacc_qp_dense_launch(q, grid, stream));
    }
    eepacc::eepacc_fb_apply_args p;
    p.cfg = h->d_cfg; p.B = B; p.x = h->fb_x; p.cost = h->fb_cost; p.qp_status = h->fb_qpstat; p.meas = h->fb_meas;
    p.A22 = h->fb_A22; p.D2 = h->fb_D2; p.out = out; p.s_pred = s_pred; p.v_pred = v_pred; p.status = status;
    p.carry = mode == 1 ? h->fb_carry : nullptr;
    const bool keep_pred = h->cfg.paramEstSetting == 2;
    if (keep_pred) { p.s_pred = h->fb_sp; p.v_pred = h->fb_vp; }      // stride B: fb_sp/fb_vp hold [N+1][B]
    HIPCHK(eepacc::launch_fb_apply(p, stream));
    if (keep_pred && s_pred && v_pred) {
        HIPCHK(hipMemcpyAsync(s_pred, h->fb_sp, (size_t)B * (N + 1) * sizeof(double), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(v_pred, h->fb_vp, (size_t)B * (N + 1) * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    // the solution is the proximal centre / initial guess of the next step
    HIPCHK(hipMemcpyAsync(h->fb_x0, h->fb_x, (size_t)B * nV * sizeof(double), hipMemcpyDeviceToDevice, stream));
    h->fb_k_done += 1;
    h->last_B = B;
    return EEPACC_OK;
}

extern "C" int eepacc_fb_step(eepacc_handle* h, int B, const double* s, const double* v, const double* v_prev,
                              const double* a_prev, const double* Fm_prev, const double* Fb_prev, const double* t0,
                              const double* s_tv, const double* v_tv, const double* a_tv_prev,
                              double* out, double* s_pred, double* v_pred, int32_t* status, void* stream) {
    (void)v_prev; (void)Fm_prev; (void)Fb_prev;   // accepted and unused, as in CreateQP_FB.m:1 (inputs v_minus1, Fm_minus1, Fb_minus1)
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    if (B < 0 || B > h->max_batch) return fail(EEPACC_EINVAL, "B exceeds max_batch of the handle");
    if (B == 0) return EEPACC_OK;
    if (!s || !v || !a_prev || !t0 || !s_tv || !v_tv || !a_tv_prev || !out || !status)
        return fail(EEPACC_EINVAL, "eepacc_fb_step: NULL buffer");
    HIPCHK(hipSetDevice(h->device));
    if (h->fbs) {
        eepacc::fbs_step_args a;
        a.cfg = h->d_cfg; a.B = B; a.k_step = h->fb_k_done;
        a.s = s; a.v = v; a.a_prev = a_prev; a.t0 = t0; a.s_tv = s_tv; a.v_tv = v_tv; a.a_tv_prev = a_tv_prev;
        a.state = h->fbs_state; a.hb = h->fbs_hb; a.out = out; a.s_pred = s_pred; a.v_pred = v_pred;
        a.status = status; a.iters = h->d_iters;
        HIPCHK(eepacc::launch_fbs_step(a, h->cfg.N, (hipStream_t)stream));
        h->fb_k_done += 1; h->last_B = B; h->fb_by_step = true;
        return EEPACC_OK;
    }
    int rc = fb_prepare(h, B);
    if (rc != EEPACC_OK) return rc;
    h->fb_by_step = true;              // after fb_prepare: its first (re)allocation clears the flag
    return fb_one_step(h, B, 0, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, out, s_pred, v_pred, status, (hipStream_t)stream);
}

extern "C" int eepacc_run_fbmpc(eepacc_handle* h, int B, int n_steps, const double* s0, const double* v0,
                                const double* a_minus1, const double* s_tv, const double* v_tv,
                                double* traj, int32_t* status, void* stream) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    if (B < 0 || B > h->max_batch || n_steps < 0) return fail(EEPACC_EINVAL, "eepacc_run_fbmpc: bad B / n_steps");
    if (B == 0 || n_steps == 0) return EEPACC_OK;
    if (!s0 || !v0 || !a_minus1 || !s_tv || !v_tv || !traj || !status)
        return fail(EEPACC_EINVAL, "eepacc_run_fbmpc: NULL buffer");
    HIPCHK(hipSetDevice(h->device));
    if (h->fb_k_done > 0 && B != h->last_B)
        return fail(EEPACC_EINVAL, "eepacc_run_fbmpc: B changed while resuming; call eepacc_reset first");
    if (h->fb_k_done > 0 && h->fb_by_step)
        return fail(EEPACC_EINVAL, "eepacc_run_fbmpc after eepacc_fb_step: the per-step operator keeps no closed-loop state to resume from; call eepacc_reset first");
    if (h->fbs) {
        // work-unit length: 16 MPC steps, shorter for short launches so that every resident wave still gets several units
        int chunk_steps = eepacc::pick_chunk_steps(n_steps, B, h->num_cus * 7);
        eepacc::fbs_run_args a;
        a.cfg = h->d_cfg; a.B = B; a.k_start = h->fb_k_done; a.n_steps = n_steps;
        a.s0 = s0; a.v0 = v0; a.a_m1 = a_minus1; a.s_tv = s_tv; a.v_tv = v_tv;
        a.carry = h->fbs_carry; a.state = h->fbs_state; a.hb = h->fbs_hb; a.traj = traj; a.status = status;
        a.iters_total = h->d_iters; a.work_counter = h->d_counter; a.done = h->d_done; a.err_word = h->d_err;
        a.chunk_steps = chunk_steps; a.spin_limit = 1 << 26;
        if (const char* ev = getenv("EEPACC_DEBUG_SPIN_LIMIT")) a.spin_limit = atoi(ev);
        a.cold = 0;
        if (const char* ev = getenv("EEPACC_DEBUG_FBS_COLD")) a.cold = atoi(ev) != 0;
        HIPCHK(eepacc::launch_fbs_run(a, h->cfg.N, h->num_cus, (hipStream_t)stream));
        h->fb_k_done += n_steps; h->last_B = B;
        return EEPACC_OK;
    }
    int rc = fb_prepare(h, B);
    if (rc != EEPACC_OK) return rc;
    for (int kk = 0; kk < n_steps; ++kk) {
        rc = fb_one_step(h, B, 1, s0, v0, a_minus1, nullptr, s_tv + (size_t)kk * B, v_tv + (size_t)kk * B, nullptr,
                         traj + (size_t)kk * EEPACC_OUT_N * B, nullptr, nullptr, status + (size_t)kk * B,
                         (hipStream_t)stream);
        if (rc != EEPACC_OK) return rc;
    }
    return EEPACC_OK;
}

// Host-pointer wrappers (what the MEX gateways mex/RunOpt_*MPC.c call): copy in, reset, run, wait, copy out.
namespace {
struct DevBufs {       // device scratch of the wrappers, freed on every return path
    double *in = nullptr, *tv = nullptr, *traj = nullptr;
    int32_t* status = nullptr;
    ~DevBufs() {
        if (in) (void)hipFree(in);
        if (tv) (void)hipFree(tv);
        if (traj) (void)hipFree(traj);
        if (status) (void)hipFree(status);
    }
};
}  // namespace

static int run_host(eepacc_handle* h, bool fb, int B, int n_steps, const double* s0, const double* v0,
                    const double* a_minus1, const double* s_tv, const double* v_tv, double* traj, int32_t* status) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    if (B < 1 || B > h->max_batch || n_steps < 1) return fail(EEPACC_EINVAL, "bad B / n_steps");
    if (!s0 || !v0 || !a_minus1 || !s_tv || !v_tv || !traj || !status) return fail(EEPACC_EINVAL, "NULL buffer");
    HIPCHK(hipSetDevice(h->device));
    DevBufs d;
    const size_t nB = (size_t)B, nT = (size_t)n_steps * B;
    HIPCHK(hipMalloc(&d.in, 3 * nB * sizeof(double)));
    HIPCHK(hipMalloc(&d.tv, 2 * nT * sizeof(double)));
    HIPCHK(hipMalloc(&d.traj, nT * EEPACC_OUT_N * sizeof(double)));
    HIPCHK(hipMalloc(&d.status, nT * sizeof(int32_t)));
    HIPCHK(hipMemcpy(d.in, s0, nB * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.in + nB, v0, nB * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.in + 2 * nB, a_minus1, nB * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.tv, s_tv, nT * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.tv + nT, v_tv, nT * sizeof(double), hipMemcpyHostToDevice));
    int rc = eepacc_reset(h);
    if (rc != EEPACC_OK) return rc;
    rc = fb ? eepacc_run_fbmpc(h, B, n_steps, d.in, d.in + nB, d.in + 2 * nB, d.tv, d.tv + nT, d.traj, d.status, nullptr)
            : eepacc_run_abmpc(h, B, n_steps, d.in, d.in + nB, d.in + 2 * nB, d.tv, d.tv + nT, d.traj, d.status, nullptr);
    if (rc != EEPACC_OK) return rc;
    rc = eepacc_synchronize(h, nullptr);
    HIPCHK(hipMemcpy(traj, d.traj, nT * EEPACC_OUT_N * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, d.status, nT * sizeof(int32_t), hipMemcpyDeviceToHost));
    return rc;
}

// RunOpt_BLMPC by name: the ABMPC entry points on a handle that was created as the baseline controller
static int need_bl(const eepacc_handle* h) {
    if (!h) return fail(EEPACC_EINVAL, "NULL handle");
    if (!h->cfg.bl_mode) return fail(EEPACC_EINVAL, "this handle was not created with bl_mode = 1 (RunOpt_BLMPC)");
    return EEPACC_OK;
}
extern "C" int eepacc_bl_step(eepacc_handle* h, int B, const double* s, const double* v, const double* a_prev,
                              const double* t0, const double* s_tv, const double* v_tv, const double* a_tv_prev,
                              double* out, double* s_pred, double* v_pred, int32_t* status, void* stream) {
    const int rc = need_bl(h);
    return rc != EEPACC_OK ? rc : eepacc_ab_step(h, B, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, out, s_pred, v_pred, status, stream);
}
extern "C" int eepacc_run_blmpc(eepacc_handle* h, int B, int n_steps, const double* s0, const double* v0,
                                const double* a_minus1, const double* s_tv, const double* v_tv, double* traj,
                                int32_t* status, void* stream) {
    const int rc = need_bl(h);
    return rc != EEPACC_OK ? rc : eepacc_run_abmpc(h, B, n_steps, s0, v0, a_minus1, s_tv, v_tv, traj, status, stream);
}
extern "C" int eepacc_run_blmpc_host(eepacc_handle* h, int B, int n_steps, const double* s0, const double* v0,
                                     const double* a_minus1, const double* s_tv, const double* v_tv,
                                     double* traj, int32_t* status) {
    const int rc = need_bl(h);
    return rc != EEPACC_OK ? rc : run_host(h, false, B, n_steps, s0, v0, a_minus1, s_tv, v_tv, traj, status);
}

extern "C" int eepacc_run_abmpc_host(eepacc_handle* h, int B, int n_steps, const double* s0, const double* v0,
                                     const double* a_minus1, const double* s_tv, const double* v_tv,
                                     double* traj, int32_t* status) {
    return run_host(h, false, B, n_steps, s0, v0, a_minus1, s_tv, v_tv, traj, status);
}

extern "C" int eepacc_run_fbmpc_host(eepacc_handle* h, int B, int n_steps, const double* s0, const double* v0,
                                     const double* a_minus1, const double* s_tv, const double* v_tv,
                                     double* traj, int32_t* status) {
    return run_host(h, true, B, n_steps, s0, v0, a_minus1, s_tv, v_tv, traj, status);
}
