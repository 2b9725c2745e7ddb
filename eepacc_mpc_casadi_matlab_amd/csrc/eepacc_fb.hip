// eepacc_fb.hip -- FBMPC per-step pipeline around the dense QP
// operator: measurement block + estimators + bounds + CreateQP_FB + condensing in one kernel
// (one workgroup per instance), then eepacc_qp_dense.hip, then extraction / plant carry.
//
// Reference: ABO/RunOpt_FBMPC.m:161-331 (loop), ABO/Functions/MPCs/CreateQP_FB.m:158-489 (QP),
// ABO/Functions/MPCs/TransformToDenseFormulation.m:30-91 (condensing).  The condensing is not
// done as literal dense products: with B_k = T_k/(lambda m) [0 0; 1 1] only the state rows of Psi
// are dense, and they are the columns Sv[.,i], Ss[.,i] of the speed / position sensitivities to
// the force of stage i; the sparse-form Hessian is block tridiagonal in (v,Fm,Fb)_k.  Each
// entry of the dense H, g, A is assembled directly from those pieces (O(N) work per entry).
//
// Dense variable order per stage (6): Fm, Fb, xi_v, xi_h, xi_s, xi_f.  Rows per stage: 26, in the
// order of CreateQP_FB.m:311-473, plus the two terminal rows :478-489.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "eepacc_fb.h"
#include "eepacc_stage.h"
#include "../../include/eepacc.h"

namespace eepacc {
namespace {

constexpr int FT = 256;
constexpr int kRowsPerStage = 26;

struct RowDesc {
    double as, av, aFm, aFb, aFmp, aFbp, sc;   // coefficients; sc = slack coefficient
    int slack;                                  // 0..3 = xi_v, xi_h, xi_s, xi_f ; -1 none
    double lb, ub;
};

struct StageData {      // per-stage scalars in LDS
    double *s_est, *v_est, *stv_est, *v_lim, *v_curv, *v_stop, *v_TL, *a_min, *a_max, *j_min, *j_max;
    double *A22, *D2, *ds, *dv, *Dblk, *Oblk, *cs, *tmp, *Sv, *Ss, *scal;
};

__device__ void carve(StageData& S, double* p, int N) {
    const int n1 = N + 1;
    S.s_est = p; p += n1; S.v_est = p; p += n1; S.stv_est = p; p += n1;
    S.v_lim = p; p += n1; S.v_curv = p; p += n1; S.v_stop = p; p += n1; S.v_TL = p; p += n1;
    S.a_min = p; p += n1; S.a_max = p; p += n1; S.j_min = p; p += n1; S.j_max = p; p += n1;
    S.A22 = p; p += n1; S.D2 = p; p += n1; S.ds = p; p += n1; S.dv = p; p += n1;
    S.Dblk = p; p += 9 * n1; S.Oblk = p; p += 9 * n1; S.cs = p; p += 3 * n1; S.tmp = p; p += 3 * n1;
    S.Sv = p; p += (size_t)n1 * N; S.Ss = p; p += (size_t)n1 * N; S.scal = p; p += 16;
}

// row t of stage k (k == N: terminal rows t = 0,1)
__device__ RowDesc fb_row(const DevCfg& C, const StageData& S, int k, int t, double s_0, double v_0, double a_minus1) {
    RowDesc R;
    R.as = R.av = R.aFm = R.aFb = R.aFmp = R.aFbp = R.sc = 0.0;
    R.slack = -1; R.lb = -INFINITY; R.ub = INFINITY;
    const int N = C.N;
    const double lm = C.lambda * C.m, za = C.zeta_a;
    const double zeta_rg = C.m * C.g * (C.c_r * C.cos_theta0 + C.sin_theta0);
    if (k == N) {                                           // CreateQP_FB.m:478-489
        R.as = 1.0;
        if (t == 0) R.ub = S.stv_est[N - 1] - C.h_min;
        else { R.av = C.tau_min; R.ub = S.stv_est[N - 1]; }
        return R;
    }
    const double ve = S.v_est[k], Tp = C.Tvec[k];
    const double base = za * ve * ve + zeta_rg;
    switch (t) {
    case 0: R.as = 1.0; R.lb = s_0; R.ub = C.s_goal; break;                       // :311-342
    case 1: R.av = 1.0; R.lb = 0.0; R.ub = C.v_max; break;
    case 2: R.aFm = 1.0; R.lb = -1e4; R.ub = 1e4; break;
    case 3: R.aFb = 1.0; R.lb = -1e4; R.ub = 0.0; break;
    case 4: R.slack = 0; R.sc = 1.0; R.lb = 0.0; break;
    case 5: R.slack = 1; R.sc = 1.0; R.lb = 0.0; break;
    case 6: R.slack = 2; R.sc = 1.0; R.lb = 0.0; break;
    case 7: R.slack = 3; R.sc = 1.0; R.lb = 0.0; break;
    case 8: {                                                                       // :359-366
        double cv = C.phi * C.T_m_max * C.T_m_max / 4.0 / C.P_m_max;
        R.av = -cv; R.aFm = C.eta_TF / C.phi; R.slack = 3; R.sc = 1.0; R.lb = -C.T_m_max; break; }
    case 9: {
        double cv = C.phi * C.T_m_max * C.T_m_max / 4.0 / C.P_m_max;
        R.av = cv; R.aFm = 1.0 / C.eta_TF / C.phi; R.slack = 3; R.sc = -1.0; R.ub = C.T_m_max; break; }
    case 10: {                                                                      // :369-377
        double zeta_w = C.m * C.g * (C.L_f * C.cos_theta0 + C.h_g * C.sin_theta0);
        R.aFm = C.L / C.mu + C.h_g; R.aFb = C.h_g; R.slack = 3; R.sc = 1.0; R.lb = -zeta_w + C.h_g * zeta_rg; break; }
    case 11: {
        double zeta_w = C.m * C.g * (C.L_f * C.cos_theta0 + C.h_g * C.sin_theta0);
        R.aFm = C.L / C.mu - C.h_g; R.aFb = -C.h_g; R.slack = 3; R.sc = -1.0; R.ub = zeta_w - C.h_g * zeta_rg; break; }
    case 12: R.aFm = 1.0; R.aFb = 1.0; R.slack = 3; R.sc = -1.0; R.ub = C.mu * C.m * C.g * C.cos_theta0; break;   // :380-387
    case 13: R.aFm = 1.0; R.aFb = 1.0; R.slack = 3; R.sc = 1.0; R.lb = -C.mu * C.m * C.g * C.cos_theta0; break;
    case 14: R.aFm = 1.0; R.aFb = 1.0; R.slack = 3; R.sc = -1.0; R.ub = lm * S.a_max[k] + base; break;            // :390-397
    case 15: R.aFm = 1.0; R.aFb = 1.0; R.slack = 3; R.sc = 1.0; R.lb = lm * S.a_min[k] + base; break;
    case 16: case 17: {                                                             // :400-418
        R.aFm = 1.0; R.aFb = 1.0; R.slack = 3;
        double rhs_j = (t == 16) ? S.j_max[k] : S.j_min[k], rhs;
        if (k == 0) rhs = lm * (Tp * rhs_j + a_minus1) + base;
        else {
            double vp = S.v_est[k - 1];
            R.aFmp = -1.0; R.aFbp = -1.0;
            rhs = lm * Tp * rhs_j + za * (ve * ve - vp * vp);      // + Dzeta_rg = 0: slope_est is constant
        }
        if (t == 16) { R.sc = -1.0; R.ub = rhs; } else { R.sc = 1.0; R.lb = rhs; }
        break; }
    case 18: R.av = 1.0; R.slack = 3; R.sc = -1.0; R.ub = S.v_lim[k]; break;        // :421-442
    case 19: R.av = 1.0; R.slack = 3; R.sc = -1.0; R.ub = S.v_curv[k]; break;
    case 20: R.av = 1.0; R.slack = 2; R.sc = -1.0; R.ub = S.v_stop[k]; break;
    case 21: R.av = 1.0; R.slack = 2; R.sc = -1.0; R.ub = S.v_TL[k]; break;
    case 22: R.av = 1.0; R.slack = 0; R.sc = 1.0; R.lb = fmin(S.v_lim[k], S.v_curv[k]); break;   // :445-448
    case 23: R.as = 1.0; R.slack = 2; R.sc = -1.0; R.ub = S.stv_est[k] - C.h_min; break;        // :451-458
    case 24: R.as = 1.0; R.av = C.tau_min; R.slack = 2; R.sc = -1.0; R.ub = S.stv_est[k]; break;
    default: {                                                                      // :461-473
        const double T_hwp = 2.0, A_hwp = 2.0, G_hwp = -0.0246 * T_hwp + 0.010819;
        R.as = 1.0; R.slack = 1; R.sc = -1.0;
        if (C.FBuseTaylor) { R.av = T_hwp + 2.0 * G_hwp * ve; R.ub = S.stv_est[k] - A_hwp + G_hwp * ve * ve; }
        else { R.av = T_hwp + G_hwp * ve; R.ub = S.stv_est[k] - A_hwp; }
        break; }
    }
    return R;
}

__global__ void __launch_bounds__(FT) k_fb_build(eepacc_fb_args a) {
    extern __shared__ __align__(16) double smem_d[];
    const DevCfg& C = *a.cfg;
    const int N = C.N, B = a.B, b = a.b0 + blockIdx.x, tid = threadIdx.x;
    const int nV = 6 * N, nC = C.fb_row0[N] + 2;
    StageData S;
    carve(S, smem_d, N);
    const double Ts = C.Tvec[0];
    const double lm = C.lambda * C.m, za = C.zeta_a;
    const double zeta_rg = C.m * C.g * (C.c_r * C.cos_theta0 + C.sin_theta0);
    // ---- measurement block (ABO/RunOpt_FBMPC.m:165-200)
    if (tid == 0) {
        double s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, vtvm = 0.0;
        if (a.mode == 0) {
            s = a.s[b]; v = a.v[b]; a_prev = a.a_prev[b]; t0 = a.t0[b];
            s_tv = a.s_tv[b]; v_tv = a.v_tv[b]; a_tv_prev = a.a_tv_prev[b];
        } else if (a.k_step == 0) {
            s = a.s[b]; v = a.v[b]; a_prev = a.a_prev[b]; t0 = 0.0;
            s_tv = a.s_tv[b]; v_tv = 0.0; a_tv_prev = 0.0; vtvm = 0.0;
        } else {
            const double s_prev = a.carry[0 * (size_t)B + b], v_prev = a.carry[1 * (size_t)B + b];
            const double Fm = a.carry[2 * (size_t)B + b], Fb = a.carry[3 * (size_t)B + b];
            const double vtv_prev = a.carry[4 * (size_t)B + b];
            plant_rk4(C, s_prev, v_prev, Fm + Fb, s, v);                                // :188
            a_prev = (v - v_prev) / Ts;
            t0 = a.k_step * Ts;
            s_tv = a.s_tv[b];
            vtvm = a.v_tv[b];
            v_tv = vtvm;
            a_tv_prev = (vtvm - vtv_prev) / Ts;
            S.scal[8] = v_prev;
        }
        S.scal[0] = s; S.scal[1] = v; S.scal[2] = a_prev; S.scal[3] = t0;
        S.scal[4] = s_tv; S.scal[5] = v_tv; S.scal[6] = a_tv_prev; S.scal[7] = vtvm;
    }
    __syncthreads();
    const double s_0 = S.scal[0], v_0 = S.scal[1], a_minus1 = S.scal[2], t_0 = S.scal[3];
    // ---- estimators and bounds (A2, A3)
    if (tid <= N) {
        double se, ve, st, vt;
        if (C.paramEstSetting == 2) {
            // EstimateVehicleTrajectory.m:81-88: [x_curr; prev(3:end); prev(end) + Ts v_prev(end)]
            const int idx = tid < N ? tid + 1 : N;
            const double ps = a.sp_prev[(size_t)idx * B + b], pv = a.vp_prev[(size_t)idx * B + b];
            se = tid == 0 ? s_0 : (tid < N ? ps : ps + C.Tvec[N - 1] * pv);
            ve = tid == 0 ? v_0 : pv;
        } else {
            estimate_traj(C, C.paramEstSetting, C.tConstACC_ego, s_0, v_0, a_minus1, tid, se, ve);
        }
        estimate_traj(C, C.TVestSetting, C.tConstACC_tar, S.scal[4], S.scal[5], S.scal[6], tid, st, vt);
        S.s_est[tid] = se; S.v_est[tid] = ve; S.stv_est[tid] = st;
        if (tid < N)
            route_bounds(C, se, ve, t_0, tid, S.v_lim[tid], S.v_curv[tid], S.v_stop[tid], S.v_TL[tid],
                         S.a_min[tid], S.a_max[tid], S.j_min[tid], S.j_max[tid]);
    }
    __syncthreads();
    // ---- state-space model with the A(k)/D(k) index quirk (ABO/RunOpt_FBMPC.m:78-90, 247-259)
    if (tid < N) {
        const int k = tid;
        double A22 = a.A22[(size_t)k * B + b], D2 = a.D2[(size_t)k * B + b];
        if (a.k_step == 0) {
            if (C.FBuseTaylor) { A22 = 1.0 - 2.0 * C.Tvec[k] * za * v_0 * v_0 / lm; D2 = C.Tvec[k] / lm * (za * v_0 * v_0); }
            else { A22 = 1.0; D2 = C.Tvec[k] / lm * (-za * v_0 * v_0); }
        }
        if (C.FBuseTaylor) {
            if (k == a.k_step) {
                const int i = N - 1;
                const double vi = S.v_est[i];
                A22 = 1.0 - 2.0 * C.Tvec[i] * za * vi / lm;
                D2 = C.Tvec[i] / lm * (za * vi * vi - zeta_rg);
            }
        } else {
            const double vi = S.v_est[k];
            D2 = C.Tvec[k] / lm * (-za * vi * vi - zeta_rg);
        }
        a.A22[(size_t)k * B + b] = A22; a.D2[(size_t)k * B + b] = D2;
        S.A22[k] = A22; S.D2[k] = D2;
    }
    __syncthreads();
    // ---- sensitivities of (s_k, v_k) to the force of stage i, and the free response d
    if (tid < N) {
        const int i = tid;
        double ss = 0.0, sv = 0.0;
        for (int k = 0; k <= N; ++k) {
            if (k == i + 1) { ss = 0.0; sv = C.Tvec[i] / lm; }
            else if (k > i + 1) { double ns = ss + C.Tvec[k - 1] * sv; sv = S.A22[k - 1] * sv; ss = ns; }
            S.Ss[(size_t)k * N + i] = ss; S.Sv[(size_t)k * N + i] = sv;
        }
    } else if (tid == N) {
        double ss = s_0, sv = v_0;
        S.ds[0] = ss; S.dv[0] = sv;
        for (int k = 1; k <= N; ++k) {
            double ns = ss + C.Tvec[k - 1] * sv;
            sv = S.A22[k - 1] * sv + S.D2[k - 1];
            ss = ns;
            S.ds[k] = ss; S.dv[k] = sv;
        }
    }
    // ---- block-tridiagonal sparse-form Hessian over (v,Fm,Fb)_k and linear term (CreateQP_FB.m:181-215)
    if (tid <= N) {
        const int k = tid;
        double D[9], O[9], cs[3];
        for (int e = 0; e < 9; ++e) { D[e] = 0.0; O[e] = 0.0; }
        cs[0] = cs[1] = cs[2] = 0.0;
        if (k < N) {
            const double w_P = C.fb_w[0], w_a = C.fb_w[1], w_j = C.fb_w[2];
            const double* bq = C.b_quadr;
            const double K = (30.0 / M_PI) * C.phi;
            const double ve = S.v_est[k];
            // power :181-184
            D[0] += w_P * 2.0 * K * K * bq[5]; D[1] += w_P * K * bq[4]; D[3] += w_P * K * bq[4]; D[4] += w_P * 2.0 * bq[3];
            cs[0] += w_P * K * bq[2]; cs[1] += w_P * bq[1];
            // acceleration :187-191
            const double fa = 2.0 * w_a / (lm * lm);
            D[0] += fa * (za * za * ve * ve + za * zeta_rg);
            D[1] += fa * (-za * ve); D[2] += fa * (-za * ve); D[3] += fa * (-za * ve); D[6] += fa * (-za * ve);
            D[4] += fa; D[5] += fa; D[7] += fa; D[8] += fa;
            cs[1] += w_a / (lm * lm) * (-2.0 * zeta_rg); cs[2] += w_a / (lm * lm) * (-2.0 * zeta_rg);
            // jerk :194-208
            const double Tp = C.Tvec[k];
            const double f = 2.0 * w_j / ((lm * Tp) * (lm * Tp));
            if (k == 0) {
                D[4] += f; D[5] += f; D[7] += f; D[8] += f;
                const double cc = 2.0 * w_j * (za * v_0 * v_0 + zeta_rg + lm * a_minus1) / ((lm * Tp) * (lm * Tp));
                cs[1] -= cc; cs[2] -= cc;
            } else {
                const double vk = ve, vp = S.v_est[k - 1];
                // lower-right block of the 6x6 coupling (current stage) and the coupling block O_k
                D[0] += f * (za * za * vp * vp);
                D[1] += f * (-za * vk); D[2] += f * (-za * vk); D[3] += f * (-za * vk); D[6] += f * (-za * vk);
                D[4] += f; D[5] += f; D[7] += f; D[8] += f;
                O[0] = f * (-za * za * vk * vp); O[1] = f * (za * vp); O[2] = f * (za * vp);
                O[3] = f * (za * vk); O[4] = -f; O[5] = -f;
                O[6] = f * (za * vk); O[7] = -f; O[8] = -f;
            }
            if (k + 1 < N) {
                // upper-left block of stage k+1's coupling lands on this stage
                const double Tn = C.Tvec[k + 1];
                const double fn = 2.0 * w_j / ((lm * Tn) * (lm * Tn));
                const double vk = S.v_est[k + 1], vp = ve;
                D[0] += fn * (za * za * vk * vk);
                D[1] += fn * (-za * vp); D[2] += fn * (-za * vp); D[3] += fn * (-za * vp); D[6] += fn * (-za * vp);
                D[4] += fn; D[5] += fn; D[7] += fn; D[8] += fn;
            }
        }
        for (int e = 0; e < 9; ++e) { S.Dblk[9 * k + e] = D[e]; S.Oblk[9 * k + e] = O[e]; }
        for (int e = 0; e < 3; ++e) S.cs[3 * k + e] = cs[e];
    }
    __syncthreads();
    // tmp = Hs d + cs on the (v,Fm,Fb) rows
    if (tid < 3 * N) {
        const int k = tid / 3, c = tid % 3;
        double t = S.Dblk[9 * k + 3 * c + 0] * S.dv[k] + S.cs[3 * k + c];
        if (k >= 1) t += S.Oblk[9 * k + 0 * 3 + c] * S.dv[k - 1];
        if (k + 1 < N) t += S.Oblk[9 * (k + 1) + 3 * c + 0] * S.dv[k + 1];
        S.tmp[3 * k + c] = t;
    }
    __syncthreads();
    const size_t lb_ = blockIdx.x;                         // index inside the chunk the QP arrays hold
    double* Hd = a.H + lb_ * nV * nV;
    double* gd = a.g + lb_ * nV;
    double* Ad = a.A + lb_ * nC * nV;
    // ---- g
    for (int col = tid; col < nV; col += FT) {
        const int i = col / 6, c = col % 6;
        double gv;
        if (c < 2) {
            gv = S.tmp[3 * i + 1 + c];
            for (int k = i + 1; k < N; ++k) gv += S.Sv[(size_t)k * N + i] * S.tmp[3 * k];
        } else {
            gv = (c == 2) ? C.fb_w[3] : (c == 3) ? 1e2 * C.fb_w[4] : (c == 4) ? C.fb_w[5] : C.fb_w[6];
        }
        gd[col] = gv;
    }
    // ---- H
    for (int idx = tid; idx < nV * nV; idx += FT) {
        const int ra = idx / nV, rb = idx % nV;
        const int i = ra / 6, ci = ra % 6, j = rb / 6, cj = rb % 6;
        double h = 0.0;
        if (ci < 2 && cj < 2) {
            const int k0 = (i > j ? i : j) + 1;
            for (int k = k0; k < N; ++k) h += S.Dblk[9 * k] * S.Sv[(size_t)k * N + i] * S.Sv[(size_t)k * N + j];
            for (int k = 1; k < N; ++k)
                h += S.Oblk[9 * k] * (S.Sv[(size_t)(k - 1) * N + i] * S.Sv[(size_t)k * N + j] +
                                      S.Sv[(size_t)k * N + i] * S.Sv[(size_t)(k - 1) * N + j]);
            // v-F cross terms
            h += S.Sv[(size_t)j * N + i] * S.Dblk[9 * j + 1 + cj];
            if (j >= 1) h += S.Sv[(size_t)(j - 1) * N + i] * S.Oblk[9 * j + 1 + cj];
            if (j + 1 < N) h += S.Sv[(size_t)(j + 1) * N + i] * S.Oblk[9 * (j + 1) + 3 * (1 + cj)];
            h += S.Sv[(size_t)i * N + j] * S.Dblk[9 * i + 1 + ci];
            if (i >= 1) h += S.Sv[(size_t)(i - 1) * N + j] * S.Oblk[9 * i + 1 + ci];
            if (i + 1 < N) h += S.Sv[(size_t)(i + 1) * N + j] * S.Oblk[9 * (i + 1) + 3 * (1 + ci)];
            // F-F
            if (i == j) h += S.Dblk[9 * i + 3 * (1 + ci) + 1 + cj];
            else if (j == i + 1) h += S.Oblk[9 * j + 3 * (1 + ci) + 1 + cj];
            else if (i == j + 1) h += S.Oblk[9 * i + 3 * (1 + cj) + 1 + ci];
        } else if (ra == rb && ci == 3) {
            h = 2.0 * C.fb_w[4];
        }
        Hd[idx] = h;
    }
    // ---- A (column-major nC x nV), lba, uba
    for (int r = tid; r < nC; r += FT) {
        // stage of row r (stages own 26 rows, 28 with the two blocked-move rows CreateQP_FB.m:346-356 after the bounds)
        int k = r / kRowsPerStage < N ? r / kRowsPerStage : N;
        while (k > 0 && r < C.fb_row0[k]) --k;
        while (k < N && r >= C.fb_row0[k + 1]) ++k;
        int t = r - C.fb_row0[k];
        RowDesc R;
        if (k < N && C.mb_mask[k] && (t == 8 || t == 9)) {
            R.as = R.av = R.aFm = R.aFb = R.aFmp = R.aFbp = R.sc = 0.0;
            R.slack = -1; R.lb = 0.0; R.ub = 0.0;
            if (t == 8) { R.aFmp = 1.0; R.aFm = -1.0; } else { R.aFbp = 1.0; R.aFb = -1.0; }
        } else {
            if (k < N && C.mb_mask[k] && t > 9) t -= 2;
            R = fb_row(C, S, k, t, s_0, v_0, a_minus1);
        }
        const double shift = R.as * S.ds[k] + R.av * S.dv[k];
        a.lba[lb_ * nC + r] = R.lb - shift;
        a.uba[lb_ * nC + r] = R.ub - shift;
        for (int i = 0; i < N; ++i) {
            const double st = R.as * S.Ss[(size_t)k * N + i] + R.av * S.Sv[(size_t)k * N + i];
            double cF0 = st, cF1 = st;
            if (i == k) { cF0 += R.aFm; cF1 += R.aFb; }
            if (i == k - 1) { cF0 += R.aFmp; cF1 += R.aFbp; }
            Ad[(size_t)(6 * i + 0) * nC + r] = cF0;
            Ad[(size_t)(6 * i + 1) * nC + r] = cF1;
#pragma unroll
            for (int c = 0; c < 4; ++c) Ad[(size_t)(6 * i + 2 + c) * nC + r] = (i == k && c == R.slack) ? R.sc : 0.0;
        }
    }
    if (tid == 0) {
        a.meas[0 * (size_t)B + b] = s_0;
        a.meas[1 * (size_t)B + b] = v_0;
        a.meas[2 * (size_t)B + b] = S.s_est[N] - s_0;          // DistHor (:208)
        a.meas[3 * (size_t)B + b] = S.scal[7];                 // lead speed carried to the next step
        a.meas[4 * (size_t)B + b] = (a.mode == 1 && a.k_step > 0) ? (v_0 - S.scal[8]) / Ts : 0.0;   // a_opt(k) (:316-318)
    }
}

// extraction (ABO/RunOpt_FBMPC.m:291-318) and closed-loop carry; thread per instance
__global__ void k_fb_apply(eepacc_fb_apply_args a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const DevCfg& C = *a.cfg;
    const int N = C.N, B = a.B, nV = 6 * N;
    const double* x = a.x + (size_t)b * nV;
    const double s = a.meas[0 * (size_t)B + b], v = a.meas[1 * (size_t)B + b];
    double* o = a.out;
    o[(size_t)EEPACC_OUT_S * B + b] = s;
    o[(size_t)EEPACC_OUT_V * B + b] = v;
    o[(size_t)EEPACC_OUT_FM * B + b] = x[0];
    o[(size_t)EEPACC_OUT_FB * B + b] = x[1];
    o[(size_t)EEPACC_OUT_A * B + b] = a.meas[4 * (size_t)B + b];
    o[(size_t)EEPACC_OUT_XI_V * B + b] = x[2];
    o[(size_t)EEPACC_OUT_XI_H * B + b] = x[3];
    o[(size_t)EEPACC_OUT_XI_S * B + b] = x[4];
    o[(size_t)EEPACC_OUT_XI_F * B + b] = x[5];
    o[(size_t)EEPACC_OUT_COST * B + b] = a.cost[b];
    o[(size_t)EEPACC_OUT_DISTHOR * B + b] = a.meas[2 * (size_t)B + b];
    o[(size_t)EEPACC_OUT_AQP * B + b] = 0.0;
    if (a.status) a.status[b] = a.qp_status[b];
    if (a.s_pred && a.v_pred) {
        const double lm = C.lambda * C.m;
        double ss = s, sv = v;
        a.s_pred[b] = ss; a.v_pred[b] = sv;
        for (int k = 0; k < N; ++k) {
            const double T = C.Tvec[k];
            const double ns = ss + T * sv;
            sv = a.A22[(size_t)k * B + b] * sv + T / lm * (x[6 * k] + x[6 * k + 1]) + a.D2[(size_t)k * B + b];
            ss = ns;
            a.s_pred[(size_t)(k + 1) * B + b] = ss; a.v_pred[(size_t)(k + 1) * B + b] = sv;
        }
    }
    if (a.carry) {
        a.carry[0 * (size_t)B + b] = s; a.carry[1 * (size_t)B + b] = v;
        a.carry[2 * (size_t)B + b] = x[0]; a.carry[3 * (size_t)B + b] = x[1];
        a.carry[4 * (size_t)B + b] = a.meas[3 * (size_t)B + b];
    }
}

}  // namespace

size_t fb_build_smem_bytes(int N) {
    const size_t n1 = (size_t)N + 1;
    return (15 * n1 + 18 * n1 + 6 * n1 + 2 * n1 * N + 16) * sizeof(double);
}

hipError_t launch_fb_build(const eepacc_fb_args& a, int N, hipStream_t stream) {
    const size_t smem = fb_build_smem_bytes(N);
    hipError_t e = hipFuncSetAttribute((const void*)k_fb_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_fb_build, dim3(a.nb), dim3(FT), smem, stream, a);
    return hipGetLastError();
}

hipError_t launch_fb_apply(const eepacc_fb_apply_args& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_fb_apply, dim3((a.B + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace eepacc
