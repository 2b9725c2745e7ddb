// Batched function evaluator of the full-route problem of RunOpt_NLP (include/eepacc_nlp.h).
//
// One thread per (route, interval): index = k*B + i, so every load and store is unit-stride across the
// routes of a wavefront.  Per thread: the RK4 x 4 integrator of ABO/RunOpt_NLP.m:262-278 on (s, v) and on the
// running cost, carried as first-order forward-mode jets over (v_k, theta_k, F_k = Fm_k + Fb_k) so that the
// objective gradient and the integrator's Jacobian block come out of the same pass; then the rows of
// :357-501 in the reference's order.  Roofline: fp64 vector ALU (a few kflop against 0.5 KB per thread).
// The objective is reduced per route by a second kernel in a fixed order (bit-reproducible).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/eepacc_nlp.h"

namespace eepacc { int set_error(int code, const std::string& msg); }

namespace {

struct NlpDev {
    int N, n_tl, flat, R, has_goal;
    double Ts, W[7], b[21], s_goal, h_min, tau_min, alpha, Fm_min, Fm_max;
    eepacc_vehicle V;
    int n_vlim, n_curv, n_slope, n_stop, n_vinc;
    int o_vlim, o_curv, o_slope, o_stop, o_vinc, o_tls, o_tlstate;   // offsets into the table blob (knots, then values)
    double tl_v[3];
};

struct Tab { const double* x; const double* y; int n; };

__device__ __forceinline__ void pwa(const Tab t, double x, double& val, double& slope) {
    int i = 0;
    for (int q = 1; q < t.n - 1; ++q) i = (t.x[q] <= x) ? q : i;     // last knot <= x, clipped to [0, n-2]
    const double dx = t.x[i + 1] - t.x[i];
    slope = dx > 0.0 ? (t.y[i + 1] - t.y[i]) / dx : 0.0;
    val = t.y[i] + slope * (x - t.x[i]);
}

// Lookup in a table stored as a piecewise QUADRATIC: on segment i (width dx, chord slope m) the value is
// y_i + m d + c_i d (d - dx), d = x - x_i; c_i = 0 gives the piecewise-linear table of casadi.interpolant('LUT','linear').
// The solver's copy of the position tables has every kink rounded this way over +- eps (two extra knots per interior
// knot, a parabola between them that matches both neighbours in value and slope: C^1), because the minimiser of a route
// can sit exactly on a knot -- a node at the end of the 1 m speed-limit ramp -- where the piecewise-linear problem has no
// KKT point in the classical sense and a Newton-type iteration stalls.
struct TabQ { const double* x; const double* y; const double* c; int n; };
__device__ __forceinline__ void pwq(const TabQ t, double x, double& val, double& slope, double& curv) {
    int i = 0;
    for (int q = 1; q < t.n - 1; ++q) i = (t.x[q] <= x) ? q : i;
    const double dx = t.x[i + 1] - t.x[i], d = x - t.x[i], c = t.c[i];
    const double m = dx > 0.0 ? (t.y[i + 1] - t.y[i]) / dx : 0.0;
    slope = fma(c, 2.0 * d - dx, m);
    val = fma(d, fma(c, d - dx, m), t.y[i]);
    curv = 2.0 * c;
}
__device__ __forceinline__ TabQ tabq(const double* blob, int off, int n) { return TabQ{blob + off, blob + off + n, blob + off + 2 * n, n}; }

// The speed along the interval depends on (v_k, theta_k, F_k) only; the motor force Fm_k enters the running cost
// alone.  So the sensitivities are carried as 3-direction jets (value + d/dv_k, d/dtheta_k, d/dF_k) and the power
// polynomial contributes through its two scalar partials.
struct J3 {
    double v, g[3];
};
__device__ __forceinline__ J3 axpy(double c, const J3& a, const J3& b) {      // c*a + b
    return J3{fma(c, a.v, b.v), {fma(c, a.g[0], b.g[0]), fma(c, a.g[1], b.g[1]), fma(c, a.g[2], b.g[2])}};
}

// RunOpt_NLP.m:226-231: P(Fm, r) with its partials dP/dFm, dP/dr
__device__ __forceinline__ void p_bat(const double* b, double F, double r, double& P, double& PF, double& Pr) {
    const double F2 = F * F, F3 = F2 * F, F4 = F2 * F2, r2 = r * r, r3 = r2 * r, r4 = r2 * r2;
    P = b[0] + b[1] * F + b[2] * r + b[3] * F2 + b[4] * F * r + b[5] * r2 + b[6] * F3 + b[7] * F2 * r + b[8] * F * r2 + b[9] * r3
      + b[10] * F4 + b[11] * F3 * r + b[12] * F2 * r2 + b[13] * F * r3 + b[14] * r4
      + b[15] * F4 * F + b[16] * F4 * r + b[17] * F3 * r2 + b[18] * F2 * r3 + b[19] * F * r4 + b[20] * r4 * r;
    PF = b[1] + 2.0 * b[3] * F + b[4] * r + 3.0 * b[6] * F2 + 2.0 * b[7] * F * r + b[8] * r2 + 4.0 * b[10] * F3 + 3.0 * b[11] * F2 * r
       + 2.0 * b[12] * F * r2 + b[13] * r3 + 5.0 * b[15] * F4 + 4.0 * b[16] * F3 * r + 3.0 * b[17] * F2 * r2 + 2.0 * b[18] * F * r3 + b[19] * r4;
    Pr = b[2] + b[4] * F + 2.0 * b[5] * r + b[7] * F2 + 2.0 * b[8] * F * r + 3.0 * b[9] * r2 + b[11] * F3 + 2.0 * b[12] * F2 * r
       + 3.0 * b[13] * F * r2 + 4.0 * b[14] * r3 + b[16] * F4 + 2.0 * b[17] * F3 * r + 3.0 * b[18] * F2 * r2 + 4.0 * b[19] * F * r3 + 5.0 * b[20] * r4;
}

__global__ void __launch_bounds__(256)
k_nlp_eval(const NlpDev C, const double* __restrict__ blob, int B, const double* __restrict__ s_tv,
           const double* __restrict__ X, const double* __restrict__ U, double* __restrict__ q_stage,
           double* __restrict__ eq, double* __restrict__ ineq, double* __restrict__ gradJ, double* __restrict__ jacF) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)C.N * B) return;
    const int k = (int)(idx / B), i = (int)(idx % B);
    const eepacc_vehicle& V = C.V;
    auto Xk = [&](int kk, int c) { return X[((size_t)kk * 4 + c) * B + i]; };
    auto Uk = [&](int kk, int c) { return U[((size_t)kk * 6 + c) * B + i]; };
    const double s0 = Xk(k, 0), v0 = Xk(k, 1), th0 = Xk(k, 2), j0 = Xk(k, 3);
    const double s1n = Xk(k + 1, 0), v1n = Xk(k + 1, 1), th1 = Xk(k + 1, 2), j1 = Xk(k + 1, 3);
    const double Fm = Uk(k, 0), Fb = Uk(k, 1), xv = Uk(k, 2), xh = Uk(k, 3), xs = Uk(k, 4), xf = Uk(k, 5);
    const double Fprev = k > 0 ? Uk(k - 1, 0) + Uk(k - 1, 1) : 0.0;                      // Uk_prev = 0 at k = 0 (:337)
    const double F = Fm + Fb;
    const double lm = V.lambda * V.m, mg = V.m * V.g;
    const double kr = (30.0 / M_PI) * V.phi;

    // ---- F = RK4 x 4 of (xdot, L), theta and j frozen over the interval (:262-278) ----
    const double c0 = cos(th0), sn0 = sin(th0);
    const double grav = V.c_r * mg * c0 + mg * sn0, dgrav = -V.c_r * mg * sn0 + mg * c0;   // and d/dtheta
    const double ilm = 1.0 / lm;
    double qF = 0.0;                                                  // d(cost integral)/dFm (direct dependence)
    // one evaluation of (xdot, L) at speed vv: acceleration jet a, running cost jet l (weights w on the accumulators)
    auto f = [&](const J3& vv, double w, J3& a, J3& qacc) {
        const double k2 = -2.0 * V.zeta_a * vv.v;
        a.v = ilm * (F - V.zeta_a * vv.v * vv.v - grav);
        a.g[0] = ilm * (k2 * vv.g[0]);
        a.g[1] = ilm * (k2 * vv.g[1] - dgrav);
        a.g[2] = ilm * (1.0 + k2 * vv.g[2]);
        double P, PF, Pr;
        p_bat(C.b, Fm, kr * vv.v, P, PF, Pr);
        const double dr = C.W[0] * Pr * kr, da = 2.0 * C.W[1] * a.v;
        qacc.v = fma(w, C.W[0] * P + C.W[1] * a.v * a.v, qacc.v);
        qacc.g[0] = fma(w, dr * vv.g[0] + da * a.g[0], qacc.g[0]);
        qacc.g[1] = fma(w, dr * vv.g[1] + da * a.g[1], qacc.g[1]);
        qacc.g[2] = fma(w, dr * vv.g[2] + da * a.g[2], qacc.g[2]);
        qF = fma(w, C.W[0] * PF, qF);
    };
    const double DT = C.Ts / 4;
    J3 v{v0, {1.0, 0.0, 0.0}}, ds{0.0, {0.0, 0.0, 0.0}}, q{0.0, {0.0, 0.0, 0.0}};
    for (int m = 0; m < 4; ++m) {
        J3 a, asum, vsum;
        f(v, DT / 6, a, q);
        asum = a; vsum = v;
        const J3 v2 = axpy(DT / 2, a, v);
        f(v2, DT / 3, a, q);
        asum = axpy(2.0, a, asum); vsum = axpy(2.0, v2, vsum);
        const J3 v3 = axpy(DT / 2, a, v);
        f(v3, DT / 3, a, q);
        asum = axpy(2.0, a, asum); vsum = axpy(2.0, v3, vsum);
        const J3 v4 = axpy(DT, a, v);
        f(v4, DT / 6, a, q);
        asum = axpy(1.0, a, asum); vsum = axpy(1.0, v4, vsum);
        ds = axpy(DT / 6, vsum, ds);
        v = axpy(DT / 6, asum, v);
    }
    const double* W = C.W;
    q_stage[idx] = q.v + C.Ts * (W[2] * j0 * j0 + W[3] * xv + W[4] * (xh * xh + 1e2 * xh) + W[5] * xs + W[6] * xf);   // L of :240
    if (gradJ) {
        double* g = gradJ + (size_t)k * 10 * B + i;
        g[0 * (size_t)B] = 0.0;
        g[1 * (size_t)B] = q.g[0];
        g[2 * (size_t)B] = q.g[1];
        g[3 * (size_t)B] = C.Ts * 2.0 * W[2] * j0;
        g[4 * (size_t)B] = qF + q.g[2];
        g[5 * (size_t)B] = q.g[2];
        g[6 * (size_t)B] = C.Ts * W[3];
        g[7 * (size_t)B] = C.Ts * W[4] * (2.0 * xh + 1e2);
        g[8 * (size_t)B] = C.Ts * W[5];
        g[9 * (size_t)B] = C.Ts * W[6];
    }
    if (jacF) {
        double* jf = jacF + (size_t)k * 6 * B + i;
        jf[0 * (size_t)B] = ds.g[0]; jf[1 * (size_t)B] = ds.g[1]; jf[2 * (size_t)B] = ds.g[2];
        jf[3 * (size_t)B] = v.g[0];  jf[4 * (size_t)B] = v.g[1];  jf[5 * (size_t)B] = v.g[2];
    }

    // ---- equality rows (:357-376) ----
    const Tab t_slope{blob + C.o_slope, blob + C.o_slope + C.n_slope, C.n_slope};
    double val, sl;
    double* e = eq + (size_t)k * 4 * B + i;
    e[0] = s0 + ds.v - s1n;
    e[(size_t)B] = v.v - v1n;
    if (C.flat) val = 0.0; else pwa(t_slope, s1n, val, sl);
    e[2 * (size_t)B] = th1 - val;
    const double c1 = cos(th1), sn1 = sin(th1);
    const double drag1 = V.zeta_a * v1n * v1n + V.c_r * mg * c1 + mg * sn1;
    const double drag0 = V.zeta_a * v0 * v0 + V.c_r * mg * c0 + mg * sn0;
    e[3 * (size_t)B] = j1 - (F - drag1 - Fprev + drag0) / (lm * C.Ts);

    // ---- inequality rows (:378-501), orientation value <= 0, then the bounds ----
    double* r = ineq + (size_t)k * C.R * B + i;
    int n = 0;
    auto put = [&](double x) { r[(size_t)n * B] = x; ++n; };
    const double a = (F - drag1) / lm;
    put(-(Fm * v1n + V.P_m_max / V.eta_TF + xf));
    put(Fm * v1n - V.P_m_max * V.eta_TF - xf);
    put(-(F + V.mu * mg * c1 + xf));
    put(F - V.mu * mg * c1 - xf);
    const double rear = V.h_g * V.lambda * a + V.h_g * V.zeta_a / V.m * v1n * v1n + V.g * (V.L_f * c1 + V.h_g * sn1);
    const double kF = V.L / (V.mu * V.m);
    put(-(kF * Fm + rear + xf));
    put(kF * Fm - rear - xf);
    const double iso_v[4] = {0.0, 5.0, 20.0, 25.0};                                        // :79-84
    const double iso_amin[4] = {-4.0, -4.0, -2.0, -2.0}, iso_amax[4] = {5.0, 5.0, 3.5, 3.5}, iso_j[4] = {5.0, 5.0, 2.5, 2.5};
    pwa(Tab{iso_v, iso_amin, 4}, v1n, val, sl);
    put(-(a - val + xf));
    pwa(Tab{iso_v, iso_amax, 4}, v1n, val, sl);
    put(a - val - xf);
    pwa(Tab{iso_v, iso_j, 4}, v1n, val, sl);
    put(-(j1 + val + xf));
    put(j1 - val - xf);
    pwa(Tab{blob + C.o_vlim, blob + C.o_vlim + C.n_vlim, C.n_vlim}, s1n, val, sl);
    put(v1n - val - xf);
    pwa(Tab{blob + C.o_curv, blob + C.o_curv + C.n_curv, C.n_curv}, s1n, val, sl);
    put(v1n - C.alpha * pow(fabs(val), -1.0 / 3.0) - xf);
    pwa(Tab{blob + C.o_stop, blob + C.o_stop + C.n_stop, C.n_stop}, s1n, val, sl);
    put(v1n - val - xs);
    for (int t = 0; t < C.n_tl; ++t) {
        pwa(Tab{blob + C.o_tls + 3 * t, C.tl_v, 3}, s1n, val, sl);
        const double tt = blob[C.o_tlstate + (size_t)t * C.N + k];
        put(v1n - val - tt - xs);
        put(-(v1n + val + 1e3 - 10.0 - tt + xs));
    }
    pwa(Tab{blob + C.o_vinc, blob + C.o_vinc + C.n_vinc, C.n_vinc}, s1n, val, sl);
    put(-(v1n - val + xv));
    const double stv = s_tv[(size_t)k * B + i];
    put(s1n - (stv - C.h_min));
    put(s1n + C.tau_min * v1n - xs - stv);
    const double T_hwp = 2.0, A_hwp = 2.0, G_hwp = -0.0246 * T_hwp + 0.010819;             // :494-496
    put(s1n + v1n * T_hwp + v1n * v1n * G_hwp - xh - (stv - A_hwp));
    put(C.Fm_min - Fm);
    put(Fm - C.Fm_max);
    put(Fb);
    put(-xv); put(-xh); put(-xs); put(-xf);
    put(-s1n);
    put(-v1n);
    put(v1n - V.v_max);
    if (C.has_goal) put(s1n - C.s_goal);
}

// objective per route: 64 routes x 16 interval slices per workgroup (loads stay unit-stride across the routes), each
// slice sums its intervals k = slice, slice + 16, ... in order, the 16 partial sums are added in slice order: a fixed
// summation tree, bit-reproducible
constexpr int SUM_SLICES = 16;
__global__ void __launch_bounds__(64 * SUM_SLICES)
k_nlp_sum(int N, int B, const double* __restrict__ q_stage, double* __restrict__ J) {
    __shared__ double part[SUM_SLICES][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double acc = 0.0;
    if (i < B)
        for (int k = slice; k < N; k += SUM_SLICES) acc += q_stage[(size_t)k * B + i];
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && i < B) {
        double t = part[0][lane];
        for (int q = 1; q < SUM_SLICES; ++q) t += part[q][lane];
        J[i] = t;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Riccati sweep of one Newton system per route (include/eepacc_nlp.h: eepacc_nlp_riccati).  One wavefront per route.
// Per stage: M = Q + AB' P AB and m = q + AB'(P c + p) entry-parallel over the lanes through LDS; the 6 x 6 control
// block is factorised redundantly in every lane (registers, no communication), lanes 0..4 back-substitute one
// right-hand side each (the four state columns and the gradient); P, p of the stage entry-parallel again.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RX = 4, RU = 6, RY = 10;
struct RicScale { double s[RU]; };

__global__ void __launch_bounds__(64)
k_riccati(int B, int N, const double* __restrict__ Qg, const double* __restrict__ qg, const double* __restrict__ ABg,
          const double* __restrict__ cg, const double* __restrict__ regg, const RicScale sc, double* __restrict__ dchi,
          double* __restrict__ du, double* __restrict__ nu, double* __restrict__ work, int32_t* __restrict__ status,
          double* __restrict__ gnorm, const double* __restrict__ qlam, const int32_t* __restrict__ rstate, int rbits) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= B) return;
    if (rstate && !((rbits >> rstate[r]) & 1)) return;            // route not in a state that needs this sweep (eepacc_nlp_solve)
    __shared__ double M[RY][RY], m[RY], AB[RX][RY], W[RX][RY], P[RX][RX], pv[RX], pc[RX], cvec[RX], Kk[RU][RX + 1], ql[RY], wad[RX];
    const double reg = regg[r];
    const double* Qr = Qg + (size_t)r * N * 100;
    const double* qr = qg + (size_t)r * N * 10;
    const double* ABr = ABg + (size_t)r * N * 40;
    const double* cr = cg + (size_t)r * N * 4;
    double* wr = work + (size_t)r * N * 50;
    if (lane < 16) P[lane >> 2][lane & 3] = 0.0;
    if (lane < 4) { pv[lane] = 0.0; wad[lane] = 0.0; }
    const double* qlr = qlam ? qlam + (size_t)r * N * 10 : nullptr;
    int bad = 0;
    double gmax = 0.0;                                     // max |reduced gradient| over the stages (lanes 0..5: one control each)
    __syncthreads();
    // stage data travel global -> registers -> LDS one stage ahead of the arithmetic (the sweep is a chain of dependent
    // stages: a load issued at the top of its own stage would put the whole memory latency on the critical path)
    double pre0, pre1 = 0.0, pre2 = 0.0;
    auto prefetch = [&](int k) {
        pre0 = Qr[(size_t)k * 100 + lane];
        if (lane < 36) pre1 = Qr[(size_t)k * 100 + 64 + lane];
        if (lane < 40) pre2 = ABr[(size_t)k * 40 + lane];
        else if (lane < 50) pre2 = qr[(size_t)k * 10 + lane - 40];
        else if (lane < 54) pre2 = cr[(size_t)k * 4 + lane - 50];
        else if (qlr) pre2 = qlr[(size_t)k * 10 + lane - 54];
    };
    prefetch(N - 1);
    for (int k = N - 1; k >= 0; --k) {
        M[lane / 10][lane % 10] = pre0;
        if (lane < 36) M[(64 + lane) / 10][(64 + lane) % 10] = pre1;
        if (lane < 40) AB[lane / 10][lane % 10] = pre2;
        else if (lane < 50) m[lane - 40] = pre2;
        else if (lane < 54) cvec[lane - 50] = pre2;
        else ql[lane - 54] = pre2;
        __syncthreads();
        if (k > 0) prefetch(k - 1);
        // dual residual with the current multipliers: adjoint w of the dynamics, reduced gradient per control
        double wn = 0.0;
        if (qlr) {
            if (lane < RU) gmax = fmax(gmax, fabs(ql[RX + lane] + AB[0][RX + lane] * wad[0] + AB[1][RX + lane] * wad[1] + AB[2][RX + lane] * wad[2] + AB[3][RX + lane] * wad[3]));
            else if (lane >= 8 && lane < 12) { const int x2 = lane - 8; wn = ql[x2] + AB[0][x2] * wad[0] + AB[1][x2] * wad[1] + AB[2][x2] * wad[2] + AB[3][x2] * wad[3]; }
        }
        if (lane < 40) {                                   // W = P AB
            const int x = lane / 10, j = lane % 10;
            W[x][j] = P[x][0] * AB[0][j] + P[x][1] * AB[1][j] + P[x][2] * AB[2][j] + P[x][3] * AB[3][j];
        } else if (lane < 44) {                            // pc = P c + p
            const int x = lane - 40;
            pc[x] = P[x][0] * cvec[0] + P[x][1] * cvec[1] + P[x][2] * cvec[2] + P[x][3] * cvec[3] + pv[x];
        }
        __syncthreads();
        for (int e = lane; e < 100; e += 64) {             // M = Q + AB' W (+ Levenberg term on the controls)
            const int i = e / 10, j = e % 10;
            double acc = M[i][j] + AB[0][i] * W[0][j] + AB[1][i] * W[1][j] + AB[2][i] * W[2][j] + AB[3][i] * W[3][j];
            if (i == j && i >= RX) acc += reg * sc.s[i - RX];
            M[i][j] = acc;
        }
        if (lane < 10) m[lane] += AB[0][lane] * pc[0] + AB[1][lane] * pc[1] + AB[2][lane] * pc[2] + AB[3][lane] * pc[3];
        __syncthreads();
        if (!qlr && lane < RU) gmax = fmax(gmax, fabs(m[RX + lane]));
        if (qlr && lane >= 8 && lane < 12) wad[lane - 8] = wn;
        // Cholesky of the control block, redundantly per lane; per-pivot relative test
        double L[RU][RU], invd[RU];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < RU; ++i) {
            double d = M[RX + i][RX + i];
            const double d0 = d;
#pragma unroll
            for (int t = 0; t < i; ++t) d -= L[i][t] * L[i][t];
            if (!(d > 1e-10 * fabs(d0))) { ok = false; d = 1.0; }
            const double li = sqrt(d), inv = 1.0 / li;
            L[i][i] = li;
            invd[i] = inv;
#pragma unroll
            for (int j2 = i + 1; j2 < RU; ++j2) {
                double v = 0.5 * (M[RX + j2][RX + i] + M[RX + i][RX + j2]);
#pragma unroll
                for (int t = 0; t < i; ++t) v -= L[j2][t] * L[i][t];
                L[j2][i] = v * inv;
            }
        }
        if (!ok && bad == 0) bad = 1 + (N - 1 - k);
        if (lane < RX + 1) {                               // K = -Muu^-1 Mux (lanes 0..3), kf = -Muu^-1 m_u (lane 4)
            double y[RU];
#pragma unroll
            for (int i = 0; i < RU; ++i) {
                double v = lane < RX ? M[RX + i][lane] : m[RX + i];
#pragma unroll
                for (int t = 0; t < i; ++t) v -= L[i][t] * y[t];
                y[i] = v * invd[i];
            }
#pragma unroll
            for (int i = RU - 1; i >= 0; --i) {
                double v = y[i];
#pragma unroll
                for (int t = i + 1; t < RU; ++t) v -= L[t][i] * y[t];
                y[i] = v * invd[i];
            }
#pragma unroll
            for (int i = 0; i < RU; ++i) Kk[i][lane] = -y[i];
        }
        __syncthreads();
        // P = sym(Mxx + Mxu K), p = m_x + Mxu kf; gains and value function of the stage -> work (30 + 16 + 4)
        double Pn = 0.0, pn = 0.0;
        if (lane < 16) {
            const int a = lane >> 2, b = lane & 3;
            double t1 = M[a][b], t2 = M[b][a];
#pragma unroll
            for (int t = 0; t < RU; ++t) { t1 += M[a][RX + t] * Kk[t][b]; t2 += M[b][RX + t] * Kk[t][a]; }
            Pn = 0.5 * (t1 + t2);
        } else if (lane < 20) {
            const int a = lane - 16;
            pn = m[a];
#pragma unroll
            for (int t = 0; t < RU; ++t) pn += M[a][RX + t] * Kk[t][RX];
        }
        if (lane < 30) wr[(size_t)k * 50 + lane] = Kk[lane / 5][lane % 5];
        __syncthreads();
        if (lane < 16) { P[lane >> 2][lane & 3] = Pn; wr[(size_t)k * 50 + 30 + lane] = Pn; }
        else if (lane < 20) { pv[lane - 16] = pn; wr[(size_t)k * 50 + 46 + lane - 16] = pn; }
    }
    __syncthreads();
    // forward sweep: lanes 0..3 carry dchi, lanes 0..5 compute du through LDS
    __shared__ double x[RX], uu[RU];
    double* dcr = dchi + (size_t)r * (N + 1) * 4;
    double* dur = du + (size_t)r * N * 6;
    double* nur = nu + (size_t)r * (N + 1) * 4;
    if (lane < RX) { x[lane] = 0.0; dcr[lane] = 0.0; nur[lane] = 0.0; nur[(size_t)N * 4 + lane] = 0.0; }
    __syncthreads();
    __shared__ double Kf[50];                              // the stage's work block: K | kf (30), P (16), p (4)
    double f0 = 0.0, f1 = 0.0;
    auto prefetch_f = [&](int k) {
        if (lane < 40) f0 = ABr[(size_t)k * 40 + lane];
        else if (lane < 44) f0 = cr[(size_t)k * 4 + lane - 40];
        if (lane < 50) f1 = wr[(size_t)k * 50 + lane];
    };
    prefetch_f(0);
    for (int k = 0; k < N; ++k) {
        if (lane < 40) AB[lane / 10][lane % 10] = f0;
        else if (lane < 44) cvec[lane - 40] = f0;
        if (lane < 50) Kf[lane] = f1;
        __syncthreads();
        if (k + 1 < N) prefetch_f(k + 1);
        if (lane < RU) {                                   // du_k = K_k dchi_k + kf_k
            const double* Kr = Kf + lane * 5;
            const double v = Kr[0] * x[0] + Kr[1] * x[1] + Kr[2] * x[2] + Kr[3] * x[3] + Kr[4];
            uu[lane] = v;
            dur[(size_t)k * 6 + lane] = v;
        } else if (lane >= 8 && lane < 12 && k > 0) {      // costate nu_k = P_k dchi_k + p_k
            const int a2 = lane - 8;
            const double* Pk = Kf + 30;
            nur[(size_t)k * 4 + a2] = Pk[a2 * 4 + 0] * x[0] + Pk[a2 * 4 + 1] * x[1] + Pk[a2 * 4 + 2] * x[2] + Pk[a2 * 4 + 3] * x[3] + Pk[16 + a2];
        }
        __syncthreads();
        double xn = 0.0;
        if (lane < RX) {
            xn = cvec[lane];
#pragma unroll
            for (int t = 0; t < RX; ++t) xn += AB[lane][t] * x[t];
#pragma unroll
            for (int t = 0; t < RU; ++t) xn += AB[lane][RX + t] * uu[t];
        }
        __syncthreads();
        if (lane < RX) { x[lane] = xn; dcr[(size_t)(k + 1) * 4 + lane] = xn; }
        __syncthreads();
    }
    if (lane == 0) status[r] = bad;
    if (gnorm) {
        __shared__ double gm[RU];
        if (lane < RU) gm[lane] = gmax;
        __syncthreads();
        if (lane == 0) gnorm[r] = fmax(fmax(fmax(gm[0], gm[1]), fmax(gm[2], gm[3])), fmax(gm[4], gm[5]));
    }
}



// Every inequality row of one interval in y = (chi_{k+1}, u_k) coordinates (0 s, 1 v, 2 p, 3 j, 4 Fm, 5 Fb, 6..9 slacks),
// orientation value <= 0, in the order of RunOpt_NLP.m:378-501 followed by the bounds: row(value, count, indices,
// gradient entries) returns the row's number, curv(row, a, b, value) reports a second-derivative entry.
template <class RowF, class CurvF>
__device__ __forceinline__ void nlp_rows(const NlpDev& C, const double* __restrict__ blob, int k, const double* x1, const double* uk,
                                         double stv, RowF row, CurvF curv) {
    const eepacc_vehicle& V = C.V;
    const double mg = V.m * V.g;
    const double Fm = uk[0], Fb = uk[1], xv = uk[2], xh = uk[3], xs = uk[4], xf = uk[5], F = Fm + Fb;
    const Tab t_slope{blob + C.o_slope, blob + C.o_slope + C.n_slope, C.n_slope};
    const double sN = x1[0], vN = x1[1], pN = x1[2], jN = x1[3];
    double thN = 0.0, dthN = 0.0;
    if (!C.flat) pwa(t_slope, sN, thN, dthN);
    const double cN = cos(thN), snN = sin(thN);
    double val, sl;
    { const int ix[3] = {1, 4, 9}; const double g[3] = {-Fm, -vN, -1.0}; const int n_last = row(-(Fm * vN + V.P_m_max / V.eta_TF + xf), 3, ix, g); curv(n_last, 1, 4, -1.0); }
    { const int ix[3] = {1, 4, 9}; const double g[3] = {Fm, vN, -1.0}; const int n_last = row(Fm * vN - V.P_m_max * V.eta_TF - xf, 3, ix, g); curv(n_last, 1, 4, 1.0); }
    { const int ix[4] = {4, 5, 0, 9}; const double g[4] = {-1.0, -1.0, V.mu * mg * snN * dthN, -1.0}; const int n_last = row(-(F + V.mu * mg * cN + xf), 4, ix, g); }
    { const int ix[4] = {4, 5, 0, 9}; const double g[4] = {1.0, 1.0, V.mu * mg * snN * dthN, -1.0}; const int n_last = row(F - V.mu * mg * cN - xf, 4, ix, g); }
    const double kF = V.L / (V.mu * V.m), kz = V.h_g * V.zeta_a / V.m;
    const double rear = V.h_g * V.lambda * pN + kz * vN * vN + V.g * (V.L_f * cN + V.h_g * snN);
    const double drear = V.g * (-V.L_f * snN + V.h_g * cN) * dthN;
    { const int ix[5] = {4, 2, 1, 0, 9}; const double g[5] = {-kF, -V.h_g * V.lambda, -2.0 * kz * vN, -drear, -1.0}; const int n_last = row(-(kF * Fm + rear + xf), 5, ix, g); curv(n_last, 1, 1, -2.0 * kz); }
    { const int ix[5] = {4, 2, 1, 0, 9}; const double g[5] = {kF, -V.h_g * V.lambda, -2.0 * kz * vN, -drear, -1.0}; const int n_last = row(kF * Fm - rear - xf, 5, ix, g); curv(n_last, 1, 1, -2.0 * kz); }
    const double iso_v[4] = {0.0, 5.0, 20.0, 25.0};
    const double iso_amin[4] = {-4.0, -4.0, -2.0, -2.0}, iso_amax[4] = {5.0, 5.0, 3.5, 3.5}, iso_j[4] = {5.0, 5.0, 2.5, 2.5};
    double cv = 0.0;                                       // curvature of a rounded kink (pwq); the ISO speed tables stay exact
    pwa(Tab{iso_v, iso_amin, 4}, vN, val, sl);
    { const int ix[3] = {2, 1, 9}; const double g[3] = {-1.0, sl, -1.0}; const int n_last = row(-(pN - val + xf), 3, ix, g); if (cv != 0.0) curv(n_last, 1, 1, cv); }
    pwa(Tab{iso_v, iso_amax, 4}, vN, val, sl);
    { const int ix[3] = {2, 1, 9}; const double g[3] = {1.0, -sl, -1.0}; const int n_last = row(pN - val - xf, 3, ix, g); if (cv != 0.0) curv(n_last, 1, 1, -cv); }
    pwa(Tab{iso_v, iso_j, 4}, vN, val, sl);
    { const int ix[3] = {3, 1, 9}; const double g[3] = {-1.0, -sl, -1.0}; const int n_last = row(-(jN + val + xf), 3, ix, g); if (cv != 0.0) curv(n_last, 1, 1, -cv); }
    { const int ix[3] = {3, 1, 9}; const double g[3] = {1.0, -sl, -1.0}; const int n_last = row(jN - val - xf, 3, ix, g); if (cv != 0.0) curv(n_last, 1, 1, -cv); }
    pwq(tabq(blob, C.o_vlim, C.n_vlim), sN, val, sl, cv);
    { const int ix[3] = {1, 0, 9}; const double g[3] = {1.0, -sl, -1.0}; const int n_last = row(vN - val - xf, 3, ix, g); if (cv != 0.0) curv(n_last, 0, 0, -cv); }
    pwq(tabq(blob, C.o_curv, C.n_curv), sN, val, sl, cv);
    { const double ac = fmax(fabs(val), 1e-300), sg = (val > 0 ? 1.0 : (val < 0 ? -1.0 : 0.0));
      const int ix[3] = {1, 0, 9}; const double g[3] = {1.0, C.alpha / 3.0 * pow(ac, -4.0 / 3.0) * sg * sl, -1.0};
      const int n_last = row(vN - C.alpha * pow(ac, -1.0 / 3.0) - xf, 3, ix, g);
      if (cv != 0.0) curv(n_last, 0, 0, C.alpha / 3.0 * pow(ac, -4.0 / 3.0) * sg * cv); }
    pwq(tabq(blob, C.o_stop, C.n_stop), sN, val, sl, cv);
    { const int ix[3] = {1, 0, 8}; const double g[3] = {1.0, -sl, -1.0}; const int n_last = row(vN - val - xs, 3, ix, g); if (cv != 0.0) curv(n_last, 0, 0, -cv); }
    for (int tl = 0; tl < C.n_tl; ++tl) {
        pwa(Tab{blob + C.o_tls + 3 * tl, C.tl_v, 3}, sN, val, sl); cv = 0.0;       // traffic-light profiles stay exact
        const double tst = blob[C.o_tlstate + (size_t)tl * C.N + k];
        { const int ix[3] = {1, 0, 8}; const double g[3] = {1.0, -sl, -1.0}; const int n_last = row(vN - val - tst - xs, 3, ix, g); if (cv != 0.0) curv(n_last, 0, 0, -cv); }
        { const int ix[3] = {1, 0, 8}; const double g[3] = {-1.0, -sl, -1.0}; const int n_last = row(-(vN + val + 1e3 - 10.0 - tst + xs), 3, ix, g); if (cv != 0.0) curv(n_last, 0, 0, -cv); }
    }
    pwq(tabq(blob, C.o_vinc, C.n_vinc), sN, val, sl, cv);
    { const int ix[3] = {1, 0, 6}; const double g[3] = {-1.0, sl, -1.0}; const int n_last = row(-(vN - val + xv), 3, ix, g); if (cv != 0.0) curv(n_last, 0, 0, cv); }
    { const int ix[1] = {0}; const double g[1] = {1.0}; const int n_last = row(sN - (stv - C.h_min), 1, ix, g); }
    { const int ix[3] = {0, 1, 8}; const double g[3] = {1.0, C.tau_min, -1.0}; const int n_last = row(sN + C.tau_min * vN - xs - stv, 3, ix, g); }
    const double T_hwp = 2.0, A_hwp = 2.0, G_hwp = -0.0246 * T_hwp + 0.010819;
    { const int ix[3] = {0, 1, 7}; const double g[3] = {1.0, T_hwp + 2.0 * G_hwp * vN, -1.0}; const int n_last = row(sN + vN * T_hwp + vN * vN * G_hwp - xh - (stv - A_hwp), 3, ix, g); curv(n_last, 1, 1, 2.0 * G_hwp); }
    { const int ix[1] = {4}; const double g[1] = {-1.0}; const int n_last = row(C.Fm_min - Fm, 1, ix, g); }
    { const int ix[1] = {4}; const double g[1] = {1.0}; const int n_last = row(Fm - C.Fm_max, 1, ix, g); }
    { const int ix[1] = {5}; const double g[1] = {1.0}; const int n_last = row(Fb, 1, ix, g); }
    for (int sidx = 6; sidx < 10; ++sidx) { const int ix[1] = {sidx}; const double g[1] = {-1.0}; const int n_last = row(-uk[sidx - 4], 1, ix, g); }
    { const int ix[1] = {0}; const double g[1] = {-1.0}; const int n_last = row(-sN, 1, ix, g); }
    { const int ix[1] = {1}; const double g[1] = {-1.0}; const int n_last = row(-vN, 1, ix, g); }
    { const int ix[1] = {1}; const double g[1] = {1.0}; const int n_last = row(vN - V.v_max, 1, ix, g); }
    if (C.has_goal) { const int ix[1] = {0}; const double g[1] = {1.0}; const int n_last = row(sN - C.s_goal, 1, ix, g); }

}

// ---------------------------------------------------------------------------------------------------------------------
// Newton-system assembly of one interior-point iteration (include/eepacc_nlp.h: eepacc_nlp_newton).  One thread per
// (route, interval).  Stage form: chi = (s, v, p, j) with p_k the acceleration at node k under the previous force and
// theta substituted; second-order forward jets over (s_k, v_k, F_k) through the RK4 x 4 integrator (the power surface
// through its scalar partials up to second order), rows with their sparse gradients and curvature, then
// Q = H_L + T'(Jr' D Jr + curvature) T,  q = grad + T'(...)  with T = [AB; 0 I]  (the stage form of DESIGN.md section 3.8).
// ---------------------------------------------------------------------------------------------------------------------
struct H3 { double v, g[3], h[6]; };                      // h: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
__device__ __forceinline__ H3 h3c(double c) { H3 r; r.v = c; for (int i = 0; i < 3; ++i) r.g[i] = 0; for (int i = 0; i < 6; ++i) r.h[i] = 0; return r; }
__device__ __forceinline__ H3 h3var(double x, int d) { H3 r = h3c(x); r.g[d] = 1.0; return r; }
__device__ __forceinline__ H3 h3axpy(double c, const H3& a, const H3& b) {
    H3 r; r.v = fma(c, a.v, b.v);
    for (int i = 0; i < 3; ++i) r.g[i] = fma(c, a.g[i], b.g[i]);
    for (int i = 0; i < 6; ++i) r.h[i] = fma(c, a.h[i], b.h[i]);
    return r;
}
__device__ __forceinline__ H3 h3mul(const H3& a, const H3& b) {
    H3 r; r.v = a.v * b.v;
    for (int i = 0; i < 3; ++i) r.g[i] = a.v * b.g[i] + b.v * a.g[i];
    const int I[6] = {0, 0, 0, 1, 1, 2}, Jx[6] = {0, 1, 2, 1, 2, 2};
    for (int e = 0; e < 6; ++e) r.h[e] = a.v * b.h[e] + b.v * a.h[e] + a.g[I[e]] * b.g[Jx[e]] + a.g[Jx[e]] * b.g[I[e]];
    return r;
}
__device__ __forceinline__ H3 h3fn(const H3& x, double f, double f1, double f2) {
    H3 r; r.v = f;
    for (int i = 0; i < 3; ++i) r.g[i] = f1 * x.g[i];
    const int I[6] = {0, 0, 0, 1, 1, 2}, Jx[6] = {0, 1, 2, 1, 2, 2};
    for (int e = 0; e < 6; ++e) r.h[e] = f1 * x.h[e] + f2 * x.g[I[e]] * x.g[Jx[e]];
    return r;
}
// second-order partials of the power surface
__device__ __forceinline__ void p_bat2(const double* b, double F, double r, double& PF, double& Pr, double& PFF, double& PFr, double& Prr) {
    const double F2 = F * F, F3 = F2 * F, F4 = F2 * F2, r2 = r * r, r3 = r2 * r, r4 = r2 * r2;
    PF = b[1] + 2.0 * b[3] * F + b[4] * r + 3.0 * b[6] * F2 + 2.0 * b[7] * F * r + b[8] * r2 + 4.0 * b[10] * F3 + 3.0 * b[11] * F2 * r
       + 2.0 * b[12] * F * r2 + b[13] * r3 + 5.0 * b[15] * F4 + 4.0 * b[16] * F3 * r + 3.0 * b[17] * F2 * r2 + 2.0 * b[18] * F * r3 + b[19] * r4;
    Pr = b[2] + b[4] * F + 2.0 * b[5] * r + b[7] * F2 + 2.0 * b[8] * F * r + 3.0 * b[9] * r2 + b[11] * F3 + 2.0 * b[12] * F2 * r
       + 3.0 * b[13] * F * r2 + 4.0 * b[14] * r3 + b[16] * F4 + 2.0 * b[17] * F3 * r + 3.0 * b[18] * F2 * r2 + 4.0 * b[19] * F * r3 + 5.0 * b[20] * r4;
    PFF = 2.0 * b[3] + 6.0 * b[6] * F + 2.0 * b[7] * r + 12.0 * b[10] * F2 + 6.0 * b[11] * F * r + 2.0 * b[12] * r2 + 20.0 * b[15] * F3
        + 12.0 * b[16] * F2 * r + 6.0 * b[17] * F * r2 + 2.0 * b[18] * r3;
    PFr = b[4] + 2.0 * b[7] * F + 2.0 * b[8] * r + 3.0 * b[11] * F2 + 4.0 * b[12] * F * r + 3.0 * b[13] * r2 + 4.0 * b[16] * F3
        + 6.0 * b[17] * F2 * r + 6.0 * b[18] * F * r2 + 4.0 * b[19] * r3;
    Prr = 2.0 * b[5] + 2.0 * b[8] * F + 6.0 * b[9] * r + 2.0 * b[12] * F2 + 6.0 * b[13] * F * r + 12.0 * b[14] * r2 + 2.0 * b[17] * F3
        + 6.0 * b[18] * F2 * r + 12.0 * b[19] * F * r2 + 20.0 * b[20] * r3;
}

__global__ void __launch_bounds__(64)
k_nlp_newton(const NlpDev C, const double* __restrict__ blob, int B, const double* __restrict__ mu_arr, double sigma, const double* __restrict__ s_tv,
             const double* __restrict__ chi, const double* __restrict__ u, const double* __restrict__ lam, const double* __restrict__ tt,
             const double* __restrict__ nu, double* __restrict__ Qo, double* __restrict__ qo, double* __restrict__ ABo,
             double* __restrict__ co, double* __restrict__ ro, double* __restrict__ qlo, const int32_t* __restrict__ rstate, int rbits) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)C.N * B) return;
    const int rt = (int)(idx / C.N), k = (int)(idx % C.N);
    if (rstate && !((rbits >> rstate[rt]) & 1)) return;
    const eepacc_vehicle& V = C.V;
    const double mu = mu_arr[rt];
    const double* x0 = chi + ((size_t)rt * (C.N + 1) + k) * 4;
    const double* x1 = x0 + 4;
    const double* uk = u + ((size_t)rt * C.N + k) * 6;
    const double* lm_ = lam + ((size_t)rt * C.N + k) * C.R;
    const double* tk = tt + ((size_t)rt * C.N + k) * C.R;
    const double* nu1 = nu + ((size_t)rt * (C.N + 1) + k + 1) * 4;
    const double s0 = x0[0], v0 = x0[1], p0 = x0[2], j0 = x0[3];
    const double Fm = uk[0], Fb = uk[1], xv = uk[2], xh = uk[3], xs = uk[4], xf = uk[5], F = Fm + Fb;
    const double lmass = V.lambda * V.m, ilm = 1.0 / lmass, mg = V.m * V.g, kr = (30.0 / M_PI) * V.phi;
    const Tab t_slope{blob + C.o_slope, blob + C.o_slope + C.n_slope, C.n_slope};
    double th0 = 0.0, dth0 = 0.0;
    if (!C.flat) pwa(t_slope, s0, th0, dth0);
    H3 th = h3c(th0); th.g[0] = dth0;
    const H3 grav = h3axpy(V.c_r * mg, h3fn(th, cos(th0), -sin(th0), -cos(th0)), h3axpy(mg, h3fn(th, sin(th0), cos(th0), -sin(th0)), h3c(0.0)));
    const H3 jF = h3var(F, 2);
    // running cost: value, gradient (s, v, F, Fm), Hessian (4 x 4 packed: 00 01 02 03 11 12 13 22 23 33)
    double qv = 0.0, qg[4] = {0, 0, 0, 0}, qh[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto f = [&](const H3& vv, double w, H3& a) {
        a = h3axpy(ilm, h3axpy(-V.zeta_a, h3mul(vv, vv), h3axpy(-1.0, grav, jF)), h3c(0.0));
        double P, PF, Pr, PFF, PFr, Prr, dummy1, dummy2;
        p_bat(C.b, Fm, kr * vv.v, P, dummy1, dummy2);
        p_bat2(C.b, Fm, kr * vv.v, PF, Pr, PFF, PFr, Prr);
        const double W0 = C.W[0], W1 = C.W[1];
        qv = fma(w, W0 * P + W1 * a.v * a.v, qv);
        for (int d = 0; d < 3; ++d) qg[d] = fma(w, W0 * Pr * kr * vv.g[d] + 2.0 * W1 * a.v * a.g[d], qg[d]);
        qg[3] = fma(w, W0 * PF, qg[3]);
        const int I[6] = {0, 0, 0, 1, 1, 2}, Jx[6] = {0, 1, 2, 1, 2, 2}, Hq[6] = {0, 1, 2, 4, 5, 7};
        for (int e = 0; e < 6; ++e)
            qh[Hq[e]] = fma(w, W0 * (Prr * kr * kr * vv.g[I[e]] * vv.g[Jx[e]] + Pr * kr * vv.h[e])
                               + 2.0 * W1 * (a.g[I[e]] * a.g[Jx[e]] + a.v * a.h[e]), qh[Hq[e]]);
        qh[3] = fma(w, W0 * PFr * kr * vv.g[0], qh[3]);
        qh[6] = fma(w, W0 * PFr * kr * vv.g[1], qh[6]);
        qh[8] = fma(w, W0 * PFr * kr * vv.g[2], qh[8]);
        qh[9] = fma(w, W0 * PFF, qh[9]);
    };
    const double DT = C.Ts / 4;
    H3 v = h3var(v0, 1), ds = h3c(0.0);
    for (int mstep = 0; mstep < 4; ++mstep) {
        H3 a, asum, vsum;
        f(v, DT / 6, a); asum = a; vsum = v;
        const H3 v2 = h3axpy(DT / 2, a, v);
        f(v2, DT / 3, a); asum = h3axpy(2.0, a, asum); vsum = h3axpy(2.0, v2, vsum);
        const H3 v3 = h3axpy(DT / 2, a, v);
        f(v3, DT / 3, a); asum = h3axpy(2.0, a, asum); vsum = h3axpy(2.0, v3, vsum);
        const H3 v4 = h3axpy(DT, a, v);
        f(v4, DT / 6, a); asum = h3axpy(1.0, a, asum); vsum = h3axpy(1.0, v4, vsum);
        ds = h3axpy(DT / 6, vsum, ds);
        v = h3axpy(DT / 6, asum, v);
    }
    H3 s1 = ds; s1.v += s0; s1.g[0] += 1.0;
    double th1 = 0.0, dth1 = 0.0;
    if (!C.flat) pwa(t_slope, s1.v, th1, dth1);
    H3 thj = h3c(th1);
    for (int i = 0; i < 3; ++i) thj.g[i] = dth1 * s1.g[i];
    for (int i = 0; i < 6; ++i) thj.h[i] = dth1 * s1.h[i];
    const H3 c1j = h3fn(thj, cos(th1), -sin(th1), -cos(th1)), s1j = h3fn(thj, sin(th1), cos(th1), -sin(th1));
    const H3 p1 = h3axpy(ilm, h3axpy(-V.zeta_a, h3mul(v, v), h3axpy(-V.c_r * mg, c1j, h3axpy(-mg, s1j, jF))), h3c(0.0));

    // ---- (chi_k, u_k) coordinates: 0 s, 1 v, 2 p, 3 j, 4 Fm, 5 Fb, 6 xi_v, 7 xi_h, 8 xi_s, 9 xi_f ----
    double AB[4][10], gl[10], Q[10][10];
    for (int i = 0; i < 10; ++i) { gl[i] = 0.0; for (int a2 = 0; a2 < 4; ++a2) AB[a2][i] = 0.0; for (int j2 = 0; j2 < 10; ++j2) Q[i][j2] = 0.0; }
    auto put_g = [&](double* row, const H3& z, double w) { row[0] += w * z.g[0]; row[1] += w * z.g[1]; row[4] += w * z.g[2]; row[5] += w * z.g[2]; };
    put_g(AB[0], s1, 1.0); put_g(AB[1], v, 1.0); put_g(AB[2], p1, 1.0); put_g(AB[3], p1, 1.0 / C.Ts);
    AB[3][2] -= 1.0 / C.Ts;
    const double cdef[4] = {s1.v - x1[0], v.v - x1[1], p1.v - x1[2], (p1.v - p0) / C.Ts - x1[3]};
    const double* W = C.W;
    gl[0] = sigma * qg[0]; gl[1] = sigma * qg[1]; gl[4] = sigma * (qg[2] + qg[3]); gl[5] = sigma * qg[2];
    gl[3] = sigma * C.Ts * 2.0 * W[2] * j0;
    gl[6] = sigma * C.Ts * W[3]; gl[7] = sigma * C.Ts * W[4] * (2.0 * xh + 1e2); gl[8] = sigma * C.Ts * W[5]; gl[9] = sigma * C.Ts * W[6];
    // Lagrangian Hessian without the barrier terms
    {
        const int dirs3[3][2] = {{0, -1}, {1, -1}, {4, 5}};                  // s, v, F -> coordinates
        const int I[6] = {0, 0, 0, 1, 1, 2}, Jx[6] = {0, 1, 2, 1, 2, 2}, Hq[6] = {0, 1, 2, 4, 5, 7};
        const double wsv = nu1[0], wv = nu1[1], wp = nu1[2] + nu1[3] / C.Ts;
        for (int e = 0; e < 6; ++e) {
            const double val = sigma * qh[Hq[e]] + wsv * s1.h[e] + wv * v.h[e] + wp * p1.h[e];
            for (int a2 = 0; a2 < 2; ++a2) for (int b2 = 0; b2 < 2; ++b2) {
                const int ia = dirs3[I[e]][a2], ib = dirs3[Jx[e]][b2];
                if (ia < 0 || ib < 0) continue;
                Q[ia][ib] += val;
                if (I[e] != Jx[e]) Q[ib][ia] += val;
            }
        }
        // Fm direction of the cost: (s,Fm) (v,Fm) (F,Fm) (Fm,Fm)
        Q[0][4] += sigma * qh[3]; Q[4][0] += sigma * qh[3];
        Q[1][4] += sigma * qh[6]; Q[4][1] += sigma * qh[6];
        Q[4][4] += 2.0 * sigma * qh[8]; Q[4][5] += sigma * qh[8]; Q[5][4] += sigma * qh[8];
        Q[4][4] += sigma * qh[9];
        Q[3][3] += sigma * C.Ts * 2.0 * W[2];
        Q[7][7] += sigma * C.Ts * 2.0 * W[4];
    }

    // ---- rows in y = (chi_{k+1}, u_k): G = Jr' D Jr + lam * curvature, gam = Jr'(mu/t + D (r + t)) ----
    double G[10][10], gam[10], gaml[10];
    for (int i = 0; i < 10; ++i) { gam[i] = 0.0; gaml[i] = 0.0; for (int j2 = 0; j2 < 10; ++j2) G[i][j2] = 0.0; }
    int n = 0;
    double* rout = ro ? ro + ((size_t)rt * C.N + k) * C.R : nullptr;
    auto row = [&](double rv, int cnt, const int* ix, const double* gv) {
        const double tv = tk[n], lv = lm_[n], D = lv / tv, wgt = mu / tv + D * (rv + tv);
        for (int a2 = 0; a2 < cnt; ++a2) {
            gam[ix[a2]] += wgt * gv[a2];
            gaml[ix[a2]] += lv * gv[a2];
            for (int b2 = 0; b2 < cnt; ++b2) G[ix[a2]][ix[b2]] += D * gv[a2] * gv[b2];
        }
        if (rout) rout[n] = rv;
        return n++;
    };
    auto curv = [&](int rowi, int a2, int b2, double val) {
        const double c2 = lm_[rowi] * val;
        G[a2][b2] += c2;
        if (a2 != b2) G[b2][a2] += c2;
    };
    nlp_rows(C, blob, k, x1, uk, s_tv[(size_t)rt * C.N + k], row, curv);

    // ---- Q = H_L + T' G T,  q = gl + T'(gam + G (c, 0)),  T = [AB; 0 I] ----
    double GA[4][10], gx[4];
    for (int a2 = 0; a2 < 4; ++a2) {
        gx[a2] = gam[a2];
        for (int b2 = 0; b2 < 4; ++b2) gx[a2] += G[a2][b2] * cdef[b2];
        for (int i = 0; i < 10; ++i) {
            double acc = 0.0;
            for (int b2 = 0; b2 < 4; ++b2) acc += G[a2][b2] * AB[b2][i];
            GA[a2][i] = acc;
        }
    }
    double* Qout = Qo + ((size_t)rt * C.N + k) * 100;
    double* qout = qo + ((size_t)rt * C.N + k) * 10;
    for (int i = 0; i < 10; ++i) {
        double qi = gl[i];
        for (int a2 = 0; a2 < 4; ++a2) qi += AB[a2][i] * gx[a2];
        if (i >= 4) { qi += gam[i]; for (int b2 = 0; b2 < 4; ++b2) qi += G[i][b2] * cdef[b2]; }
        qout[i] = qi;
        if (qlo) {                                          // gradient of the Lagrangian with the current multipliers
            double ql = gl[i];
            for (int a2 = 0; a2 < 4; ++a2) ql += AB[a2][i] * gaml[a2];
            if (i >= 4) ql += gaml[i];
            qlo[((size_t)rt * C.N + k) * 10 + i] = ql;
        }
        for (int j2 = 0; j2 < 10; ++j2) {
            double acc = Q[i][j2];
            for (int a2 = 0; a2 < 4; ++a2) acc += AB[a2][i] * GA[a2][j2];
            if (j2 >= 4) for (int a2 = 0; a2 < 4; ++a2) acc += AB[a2][i] * G[a2][j2];
            if (i >= 4) for (int a2 = 0; a2 < 4; ++a2) acc += G[i][a2] * AB[a2][j2];
            if (i >= 4 && j2 >= 4) acc += G[i][j2];
            Qout[i * 10 + j2] = acc;
        }
    }
    double* ABout = ABo + ((size_t)rt * C.N + k) * 40;
    for (int a2 = 0; a2 < 4; ++a2) for (int i = 0; i < 10; ++i) ABout[a2 * 10 + i] = AB[a2][i];
    double* cout = co + ((size_t)rt * C.N + k) * 4;
    for (int a2 = 0; a2 < 4; ++a2) cout[a2] = cdef[a2];
}


// Rows and their directional derivative along a step: r [B][N][R] and Jr dy [B][N][R] with dy = (dchi_{k+1}, du_k).
__global__ void __launch_bounds__(256)
k_nlp_rowdir(const NlpDev C, const double* __restrict__ blob, int B, const double* __restrict__ s_tv, const double* __restrict__ chi,
             const double* __restrict__ u, const double* __restrict__ dchi, const double* __restrict__ du, double* __restrict__ ro,
             double* __restrict__ jdy, const int32_t* __restrict__ rstate, int rbits) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)C.N * B) return;
    const int rt = (int)(idx / C.N), k = (int)(idx % C.N);
    if (rstate && !((rbits >> rstate[rt]) & 1)) return;
    const double* x1 = chi + ((size_t)rt * (C.N + 1) + k + 1) * 4;
    const double* uk = u + ((size_t)rt * C.N + k) * 6;
    double dy[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (dchi) {
        const double* d1 = dchi + ((size_t)rt * (C.N + 1) + k + 1) * 4;
        const double* d2 = du + ((size_t)rt * C.N + k) * 6;
        for (int i = 0; i < 4; ++i) dy[i] = d1[i];
        for (int i = 0; i < 6; ++i) dy[4 + i] = d2[i];
    }
    double* rout = ro + ((size_t)rt * C.N + k) * C.R;
    double* jout = jdy ? jdy + ((size_t)rt * C.N + k) * C.R : nullptr;
    int n = 0;
    auto row = [&](double rv, int cnt, const int* ix, const double* gv) {
        double acc = 0.0;
        for (int a2 = 0; a2 < cnt; ++a2) acc += gv[a2] * dy[ix[a2]];
        rout[n] = rv;
        if (jout) jout[n] = acc;
        return n++;
    };
    auto curv = [&](int, int, int, double) {};
    nlp_rows(C, blob, k, x1, uk, s_tv[(size_t)rt * C.N + k], row, curv);
}

// ---------------------------------------------------------------------------------------------------------------------
// Closed-loop nonlinear forward pass (include/eepacc_nlp.h: eepacc_nlp_rollout): u_k = u_base_k + alpha kf_k +
// K_k (chi_k - chi_base_k), chi_{k+1} = f(chi_k, u_k) with the RK4 x 4 integrator.  Serial in k: one wavefront per route.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_nlp_rollout(const NlpDev C, const double* __restrict__ blob, int B, const double* __restrict__ alpha,
              const double* __restrict__ chi0, const double* __restrict__ u0, const double* __restrict__ work,
              double* __restrict__ chi1, double* __restrict__ u1, const int32_t* __restrict__ rstate, int rbits) {
    // one wavefront per route: the stage's gains, base state and base controls stream global -> registers -> LDS one
    // stage ahead (as in the Riccati sweep); lanes 0..5 form the controls, lane 0 integrates the interval
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= B) return;
    if (rstate && !((rbits >> rstate[r]) & 1)) return;
    const eepacc_vehicle& V = C.V;
    const double ilm = 1.0 / (V.lambda * V.m), mg = V.m * V.g, za = V.zeta_a;
    const Tab t_slope{blob + C.o_slope, blob + C.o_slope + C.n_slope, C.n_slope};
    const double a = alpha ? alpha[r] : 0.0;
    const double* cb = chi0 + (size_t)r * (C.N + 1) * 4;
    const double* ub = u0 + (size_t)r * C.N * 6;
    const double* wk = work ? work + (size_t)r * C.N * 50 : nullptr;
    double* cn = chi1 + (size_t)r * (C.N + 1) * 4;
    double* un = u1 + (size_t)r * C.N * 6;
    __shared__ double Kf[30], cbk[4], ubk[6], x[4], uu[6];
    if (lane < 4) { x[lane] = cb[lane]; cn[lane] = cb[lane]; }
    double f0 = 0.0, f1 = 0.0;
    auto prefetch = [&](int k) {
        if (lane < 30) f1 = wk ? wk[(size_t)k * 50 + lane] : 0.0;
        if (lane >= 32 && lane < 36) f0 = cb[(size_t)k * 4 + lane - 32];
        else if (lane >= 40 && lane < 46) f0 = ub[(size_t)k * 6 + lane - 40];
    };
    prefetch(0);
    __syncthreads();
    for (int k = 0; k < C.N; ++k) {
        if (lane < 30) Kf[lane] = f1;
        if (lane >= 32 && lane < 36) cbk[lane - 32] = f0;
        else if (lane >= 40 && lane < 46) ubk[lane - 40] = f0;
        __syncthreads();
        if (k + 1 < C.N) prefetch(k + 1);
        if (lane < 6) {
            const double* Kr = Kf + lane * 5;
            const double d = a * Kr[4] + Kr[0] * (x[0] - cbk[0]) + Kr[1] * (x[1] - cbk[1]) + Kr[2] * (x[2] - cbk[2]) + Kr[3] * (x[3] - cbk[3]);
            const double v = ubk[lane] + d;
            uu[lane] = v;
            un[(size_t)k * 6 + lane] = v;
        }
        __syncthreads();
        if (lane == 0) {
            const double F = uu[0] + uu[1];
            double th = 0.0, sl;
            if (!C.flat) pwa(t_slope, x[0], th, sl);
            const double grav = V.c_r * mg * cos(th) + mg * sin(th);
            double s = x[0], v = x[1];
            const double DT = C.Ts / 4;
            for (int m2 = 0; m2 < 4; ++m2) {
                const double a1 = (F - za * v * v - grav) * ilm;
                const double v2 = v + (DT / 2) * a1;
                const double a2 = (F - za * v2 * v2 - grav) * ilm;
                const double v3 = v + (DT / 2) * a2;
                const double a3 = (F - za * v3 * v3 - grav) * ilm;
                const double v4 = v + DT * a3;
                const double a4 = (F - za * v4 * v4 - grav) * ilm;
                s = s + (DT / 6) * (v + 2 * v2 + 2 * v3 + v4);
                v = v + (DT / 6) * (a1 + 2 * a2 + 2 * a3 + a4);
            }
            double th1 = 0.0;
            if (!C.flat) pwa(t_slope, s, th1, sl);
            const double p1 = (F - za * v * v - V.c_r * mg * cos(th1) - mg * sin(th1)) * ilm;
            const double j1 = (p1 - x[2]) / C.Ts;
            x[0] = s; x[1] = v; x[2] = p1; x[3] = j1;
        }
        __syncthreads();
        if (lane < 4) cn[(size_t)(k + 1) * 4 + lane] = x[lane];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Per-route reductions of an interior-point iteration (include/eepacc_nlp.h: eepacc_nlp_steprule, eepacc_nlp_trial).
// One workgroup per route, 256 threads over the N * R rows of the route, fixed-order tree reductions in LDS.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RED_T = 256;
__device__ __forceinline__ double blk_reduce(double v, double* sh, int op) {     // op 0 sum, 1 min, 2 max
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s2 = RED_T / 2; s2 > 0; s2 >>= 1) {
        if (t < s2) {
            const double a = sh[t], b = sh[t + s2];
            sh[t] = op == 0 ? a + b : (op == 1 ? fmin(a, b) : fmax(a, b));
        }
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// out [B][8]: a_p, a_d, infeas, max |lam_new| on rows that do not hold, sum log t, e_prim, max lam t, max |lam t - mu|
__global__ void __launch_bounds__(RED_T)
k_nlp_steprule(int n, const double* __restrict__ r, const double* __restrict__ t, const double* __restrict__ lam,
               const double* __restrict__ jdy, const double* __restrict__ mu_arr, const double* __restrict__ tau_arr,
               double* __restrict__ dt, double* __restrict__ dlam, double* __restrict__ out, const int32_t* __restrict__ rstate, int rbits) {
    __shared__ double sh[RED_T];
    const int rt = blockIdx.x;
    if (rstate && !((rbits >> rstate[rt]) & 1)) return;
    const size_t base = (size_t)rt * n;
    const double mu = mu_arr[rt], tau = tau_arr[rt];
    double ap = INFINITY, ad = INFINITY, inf_ = 0.0, lami = 0.0, slog = 0.0, eprim = 0.0, c0 = 0.0, cm = 0.0;
    for (int e = threadIdx.x; e < n; e += RED_T) {
        const double rv = r[base + e], tv = t[base + e], lv = lam[base + e];
        const double rg = rv + tv;
        const bool is_i = rg > 1e-9 * (1.0 + tv);
        slog += log(tv);
        if (is_i) { inf_ += rg; eprim = fmax(eprim, rg); }
        c0 = fmax(c0, lv * tv);
        if (!jdy) ap = fmin(ap, lv * tv);                  // without a step: slot 0 carries min lam t (barrier update rule)
        cm = fmax(cm, fabs(lv * tv - mu));
        if (jdy) {
            const double D = lv / tv, jd = jdy[base + e];
            const double d = -rg - jd, ln = mu / tv + D * rg + D * jd, dl = ln - lv;
            dt[base + e] = d;
            dlam[base + e] = dl;
            if (d < 0.0) ap = fmin(ap, -tau * tv / d);
            if (dl < 0.0) ad = fmin(ad, -tau * lv / dl);
            if (is_i) lami = fmax(lami, fabs(ln));
        }
    }
    const double v0 = blk_reduce(ap, sh, 1), v1 = blk_reduce(ad, sh, 1), v2 = blk_reduce(inf_, sh, 0), v3 = blk_reduce(lami, sh, 2);
    const double v4 = blk_reduce(slog, sh, 0), v5 = blk_reduce(eprim, sh, 2), v6 = blk_reduce(c0, sh, 2), v7 = blk_reduce(cm, sh, 2);
    if (threadIdx.x == 0) {
        double* o = out + (size_t)rt * 8;
        o[0] = jdy ? fmin(v0, 1.0) : v0; o[1] = fmin(v1, 1.0); o[2] = v2; o[3] = v3; o[4] = v4; o[5] = v5; o[6] = v6; o[7] = v7;
    }
}

// trial point of the line search: slacks (rows that hold follow their row, the others keep the Newton slack unless the
// row has become satisfied beyond it), fraction-to-the-boundary test, residual and barrier sums.  out [B][3]: feasible
// (1 / 0), sum of residuals of rows that do not hold, sum log t_trial
__global__ void __launch_bounds__(RED_T)
k_nlp_trial(int n, const double* __restrict__ r, const double* __restrict__ t, const double* __restrict__ dt,
            const double* __restrict__ r_t, const double* __restrict__ a_arr, const double* __restrict__ tau_arr,
            double* __restrict__ t_t, double* __restrict__ out, const int32_t* __restrict__ rstate, int rbits) {
    __shared__ double sh[RED_T];
    const int rt = blockIdx.x;
    if (rstate && !((rbits >> rstate[rt]) & 1)) return;
    const size_t base = (size_t)rt * n;
    const double a = a_arr[rt], tau = tau_arr[rt];
    double bad = 0.0, inf_ = 0.0, slog = 0.0;
    for (int e = threadIdx.x; e < n; e += RED_T) {
        const double rv = r[base + e], tv = t[base + e], rt_ = r_t[base + e];
        const bool is_i = (rv + tv) > 1e-9 * (1.0 + tv);
        const double tn = is_i ? fmax(-rt_, tv + a * dt[base + e]) : -rt_;
        t_t[base + e] = tn;
        if (!(tn >= (1.0 - tau) * tv)) bad = 1.0;
        if (is_i) inf_ += rt_ + tn;
        slog += log(fmax(tn, 1e-300));
    }
    const double v0 = blk_reduce(bad, sh, 2), v1 = blk_reduce(inf_, sh, 0), v2 = blk_reduce(slog, sh, 0);
    if (threadIdx.x == 0) { double* o = out + (size_t)rt * 3; o[0] = v0 > 0.0 ? 0.0 : 1.0; o[1] = v1; o[2] = v2; }
}

}  // namespace

struct NlpWork;
static void nlp_work_free(NlpWork* w);

struct eepacc_nlp_handle {
    int device = 0;
    NlpDev C;
    double* d_blob = nullptr;
    double* d_q = nullptr;
    size_t q_cap = 0;
    std::vector<double> blob_host;      // the lookup tables on the host (start generator of eepacc_run_nlp_host)
    NlpDev Cr;                          // the solver's copy: position tables with their kinks rounded over +- eps_r
    double* d_blob_r = nullptr;
    double eps_r = 0.0;
    std::vector<double> tabs_host[5][2];   // the five tables as given (knots, values): source of the rounded copies
    std::vector<double> tl_s_host, tl_state_host;
    NlpWork* work = nullptr;            // device workspace of eepacc_nlp_solve (grow-only)
};

// Table blob of a handle: per table [knots | values | c] (pwq), then the traffic-light knots and phases.  eps > 0 rounds
// every interior kink of the four position tables (speed limit, curvature, stop profile, velocity incentive): knot j is
// replaced by the two knots x_j -+ e_j, e_j = min(eps, half the shorter adjoining segment), with the parabola coefficient
// (slope_right - slope_left) / (4 e_j) on the segment between them.  The slope table and eps = 0 stay exact (c = 0).
static void nlp_build_blob(const eepacc_nlp_handle* h, double eps, NlpDev& C, std::vector<double>& blob) {
    blob.clear();
    auto push = [&](int which, bool round, int& n_out) {
        const std::vector<double>& x = h->tabs_host[which][0];
        const std::vector<double>& y = h->tabs_host[which][1];
        const int n = (int)x.size();
        std::vector<double> X, Y, Cq;
        auto add = [&](double xv, double yv, double cv) { X.push_back(xv); Y.push_back(yv); Cq.push_back(cv); };
        add(x[0], y[0], 0.0);
        for (int j = 1; j + 1 < n; ++j) {
            const double gl = x[j] - x[j - 1], gr = x[j + 1] - x[j];
            if (!round || !(eps > 0.0) || !(gl > 0.0) || !(gr > 0.0)) { add(x[j], y[j], 0.0); continue; }     // a jump stays a jump
            const double sL = (y[j] - y[j - 1]) / gl, sR = (y[j + 1] - y[j]) / gr;
            if (sL == sR) { add(x[j], y[j], 0.0); continue; }
            const double e = std::fmin(eps, 0.5 * std::fmin(gl, gr));
            add(x[j] - e, y[j] - sL * e, (sR - sL) / (4.0 * e));
            add(x[j] + e, y[j] + sR * e, 0.0);
        }
        add(x[n - 1], y[n - 1], 0.0);
        const int o = (int)blob.size();
        blob.insert(blob.end(), X.begin(), X.end()); blob.insert(blob.end(), Y.begin(), Y.end()); blob.insert(blob.end(), Cq.begin(), Cq.end());
        n_out = (int)X.size();
        return o;
    };
    C.o_vlim = push(0, true, C.n_vlim);
    C.o_curv = push(1, true, C.n_curv);
    C.o_slope = push(2, false, C.n_slope);
    C.o_stop = push(3, true, C.n_stop);
    C.o_vinc = push(4, true, C.n_vinc);
    C.o_tls = (int)blob.size();
    blob.insert(blob.end(), h->tl_s_host.begin(), h->tl_s_host.end());
    C.o_tlstate = (int)blob.size();
    blob.insert(blob.end(), h->tl_state_host.begin(), h->tl_state_host.end());
}

#define NLPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return eepacc::set_error(EEPACC_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" int eepacc_nlp_sizeof_problem(void) { return (int)sizeof(eepacc_nlp_problem); }

extern "C" int eepacc_nlp_rows(const eepacc_nlp_problem* p) {
    if (!p) return EEPACC_EINVAL;
    return 17 + 2 * p->n_tl + 10 + (std::isfinite(p->s_goal) ? 1 : 0);
}

extern "C" int eepacc_nlp_create(eepacc_nlp_handle** out, const eepacc_nlp_problem* p, const eepacc_vehicle* V, int device) {
    if (!out || !p || !V) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_create: null argument");
    *out = nullptr;
    if (p->N < 1 || !(p->Ts > 0)) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_create: N >= 1 and Ts > 0 required");
    if (p->n_tl < 0 || p->n_tl > EEPACC_NLP_MAX_TL) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_create: at most 8 traffic lights");
    struct T { int n; const double* x; const double* y; const char* name; };
    const T tabs[5] = {{p->n_vlim, p->s_vlim, p->v_vlim, "speed limit"}, {p->n_curv, p->s_curv, p->curvature, "curvature"},
                       {p->n_slope, p->s_slope, p->slope, "slope"}, {p->n_stop, p->s_stop, p->v_stop, "stop"},
                       {p->n_vinc, p->s_vinc, p->v_vinc, "velocity incentive"}};
    for (const T& t : tabs) {
        if (t.n < 2 || t.n > EEPACC_NLP_MAX_KNOTS || !t.x || !t.y)
            return eepacc::set_error(EEPACC_EINVAL, std::string("eepacc_nlp_create: ") + t.name + " table needs 2..64 knots");
        for (int i = 1; i < t.n; ++i)
            if (!(t.x[i] >= t.x[i - 1])) return eepacc::set_error(EEPACC_EINVAL, std::string("eepacc_nlp_create: ") + t.name + " knots must ascend");
    }
    if (p->n_tl > 0 && (!p->tl_s || !p->tl_state)) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_create: traffic-light tables missing");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return eepacc::set_error(EEPACC_EDEVICE, "eepacc_nlp_create: no HIP device (libeepacc has no CPU path)");
    if (device < 0 || device >= ndev) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_create: bad device index");
    NLPCHK(hipSetDevice(device));
    eepacc_nlp_handle* h = new (std::nothrow) eepacc_nlp_handle();
    if (!h) return eepacc::set_error(EEPACC_ENOMEM, "eepacc_nlp_create: out of host memory");
    h->device = device;
    NlpDev& C = h->C;
    std::memset(&C, 0, sizeof(C));
    C.N = p->N; C.n_tl = p->n_tl; C.flat = p->flat ? 1 : 0; C.Ts = p->Ts;
    std::memcpy(C.W, p->W, sizeof(C.W));
    std::memcpy(C.b, p->b, sizeof(C.b));
    C.s_goal = p->s_goal; C.h_min = p->h_min; C.tau_min = p->tau_min; C.alpha = p->alpha_TTL;
    C.has_goal = std::isfinite(p->s_goal) ? 1 : 0;
    C.R = eepacc_nlp_rows(p);
    C.V = *V;
    C.Fm_min = -V->phi * V->T_m_max / V->eta_TF;                                           // RunOpt_NLP.m:194-195
    C.Fm_max = V->phi * V->T_m_max * V->eta_TF;
    std::memcpy(C.tl_v, p->tl_v, sizeof(C.tl_v));
    for (int i = 0; i < 5; ++i) { h->tabs_host[i][0].assign(tabs[i].x, tabs[i].x + tabs[i].n); h->tabs_host[i][1].assign(tabs[i].y, tabs[i].y + tabs[i].n); }
    if (p->n_tl) { h->tl_s_host.assign(p->tl_s, p->tl_s + 3 * (size_t)p->n_tl); h->tl_state_host.assign(p->tl_state, p->tl_state + (size_t)p->n_tl * p->N); }
    std::vector<double> blob;
    nlp_build_blob(h, 0.0, C, blob);                    // the exact tables: function evaluator and the single operators
    h->blob_host = blob;
    hipError_t e = hipMalloc(&h->d_blob, blob.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(h->d_blob, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        eepacc_nlp_destroy(h);
        return eepacc::set_error(EEPACC_EDEVICE, std::string("eepacc_nlp_create: ") + hipGetErrorString(e));
    }
    *out = h;
    return EEPACC_OK;
}

extern "C" void eepacc_nlp_destroy(eepacc_nlp_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_blob) (void)hipFree(h->d_blob);
    if (h->d_q) (void)hipFree(h->d_q);
    if (h->d_blob_r) (void)hipFree(h->d_blob_r);
    nlp_work_free(h->work);
    delete h;
}

extern "C" int eepacc_nlp_eval(eepacc_nlp_handle* h, int B, const double* s_tv_dev, const double* X_dev, const double* U_dev,
                               double* J_dev, double* eq_dev, double* ineq_dev, double* gradJ_dev, double* jacF_dev,
                               void* stream) {
    if (!h || B < 1 || !s_tv_dev || !X_dev || !U_dev || !J_dev || !eq_dev || !ineq_dev)
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_eval: null argument or B < 1");
    NLPCHK(hipSetDevice(h->device));
    const size_t units = (size_t)h->C.N * B;
    if (units > h->q_cap) {
        if (h->d_q) (void)hipFree(h->d_q);
        h->d_q = nullptr; h->q_cap = 0;
        NLPCHK(hipMalloc(&h->d_q, units * sizeof(double)));
        h->q_cap = units;
    }
    hipStream_t st = (hipStream_t)stream;
    const int threads = 256;
    const unsigned blocks = (unsigned)((units + threads - 1) / threads);
    hipLaunchKernelGGL(k_nlp_eval, dim3(blocks), dim3(threads), 0, st, h->C, h->d_blob, B, s_tv_dev, X_dev, U_dev, h->d_q,
                       eq_dev, ineq_dev, gradJ_dev, jacF_dev);
    NLPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_nlp_sum, dim3((B + 63) / 64), dim3(64 * SUM_SLICES), 0, st, h->C.N, B, h->d_q, J_dev);
    NLPCHK(hipGetLastError());
    return EEPACC_OK;
}

extern "C" int eepacc_nlp_synchronize(eepacc_nlp_handle* h, void* stream) {
    if (!h) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_synchronize: null handle");
    NLPCHK(hipSetDevice(h->device));
    NLPCHK(hipStreamSynchronize((hipStream_t)stream));
    return EEPACC_OK;
}

extern "C" int eepacc_nlp_riccati(int device, int B, int N, const double* Q_dev, const double* q_dev, const double* AB_dev,
                                  const double* c_dev, const double* reg_dev, const double reg_scale[6], double* dchi_dev,
                                  double* du_dev, double* nu_dev, double* work_dev, int32_t* status_dev, double* gnorm_dev,
                                  const double* qlam_dev, void* stream) {
    if (B < 1 || N < 1 || !Q_dev || !q_dev || !AB_dev || !c_dev || !reg_dev || !reg_scale || !dchi_dev || !du_dev || !nu_dev ||
        !work_dev || !status_dev)
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_riccati: null argument, B < 1 or N < 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return eepacc::set_error(EEPACC_EDEVICE, "eepacc_nlp_riccati: no HIP device (libeepacc has no CPU path)");
    if (device < 0 || device >= ndev) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_riccati: bad device index");
    NLPCHK(hipSetDevice(device));
    RicScale sc;
    for (int i = 0; i < 6; ++i) sc.s[i] = reg_scale[i];
    hipLaunchKernelGGL(k_riccati, dim3(B), dim3(64), 0, (hipStream_t)stream, B, N, Q_dev, q_dev, AB_dev, c_dev, reg_dev, sc,
                       dchi_dev, du_dev, nu_dev, work_dev, status_dev, gnorm_dev, qlam_dev, nullptr, 0);
    NLPCHK(hipGetLastError());
    return EEPACC_OK;
}

extern "C" int eepacc_nlp_newton(eepacc_nlp_handle* h, int B, const double* mu_dev, double sigma, const double* s_tv_dev, const double* chi_dev,
                                 const double* u_dev, const double* lam_dev, const double* t_dev, const double* nu_dev,
                                 double* Q_dev, double* q_dev, double* AB_dev, double* c_dev, double* rows_dev, double* qlam_dev,
                                 void* stream) {
    if (!h || B < 1 || !mu_dev || !s_tv_dev || !chi_dev || !u_dev || !lam_dev || !t_dev || !nu_dev || !Q_dev || !q_dev || !AB_dev || !c_dev)
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_newton: null argument or B < 1");
    NLPCHK(hipSetDevice(h->device));
    const size_t units = (size_t)h->C.N * B;
    hipLaunchKernelGGL(k_nlp_newton, dim3((unsigned)((units + 63) / 64)), dim3(64), 0, (hipStream_t)stream, h->C, h->d_blob, B, mu_dev, sigma,
                       s_tv_dev, chi_dev, u_dev, lam_dev, t_dev, nu_dev, Q_dev, q_dev, AB_dev, c_dev, rows_dev, qlam_dev, nullptr, 0);
    NLPCHK(hipGetLastError());
    return EEPACC_OK;
}

extern "C" int eepacc_nlp_rollout(eepacc_nlp_handle* h, int B, const double* alpha_dev, const double* chi_dev, const double* u_dev,
                                  const double* work_dev, double* chi_new_dev, double* u_new_dev, void* stream) {
    if (!h || B < 1 || !chi_dev || !u_dev || !chi_new_dev || !u_new_dev || (work_dev && !alpha_dev))
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_rollout: null argument or B < 1");
    NLPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_nlp_rollout, dim3(B), dim3(64), 0, (hipStream_t)stream, h->C, h->d_blob, B, alpha_dev, chi_dev, u_dev,
                       work_dev, chi_new_dev, u_new_dev, nullptr, 0);
    NLPCHK(hipGetLastError());
    return EEPACC_OK;
}

extern "C" int eepacc_nlp_rowdir(eepacc_nlp_handle* h, int B, const double* s_tv_dev, const double* chi_dev, const double* u_dev,
                                 const double* dchi_dev, const double* du_dev, double* rows_dev, double* jdy_dev, void* stream) {
    if (!h || B < 1 || !s_tv_dev || !chi_dev || !u_dev || !rows_dev || (jdy_dev && (!dchi_dev || !du_dev)))
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_rowdir: null argument or B < 1");
    NLPCHK(hipSetDevice(h->device));
    const size_t units = (size_t)h->C.N * B;
    hipLaunchKernelGGL(k_nlp_rowdir, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->C, h->d_blob, B, s_tv_dev,
                       chi_dev, u_dev, jdy_dev ? dchi_dev : nullptr, du_dev, rows_dev, jdy_dev, nullptr, 0);
    NLPCHK(hipGetLastError());
    return EEPACC_OK;
}

extern "C" int eepacc_nlp_steprule(int device, int B, int rows_per_route, const double* r_dev, const double* t_dev, const double* lam_dev,
                                   const double* jdy_dev, const double* mu_dev, const double* tau_dev, double* dt_dev, double* dlam_dev,
                                   double* out_dev, void* stream) {
    if (B < 1 || rows_per_route < 1 || !r_dev || !t_dev || !lam_dev || !mu_dev || !tau_dev || !out_dev || (jdy_dev && (!dt_dev || !dlam_dev)))
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_steprule: null argument");
    NLPCHK(hipSetDevice(device));
    hipLaunchKernelGGL(k_nlp_steprule, dim3(B), dim3(RED_T), 0, (hipStream_t)stream, rows_per_route, r_dev, t_dev, lam_dev, jdy_dev, mu_dev,
                       tau_dev, dt_dev, dlam_dev, out_dev, nullptr, 0);
    NLPCHK(hipGetLastError());
    return EEPACC_OK;
}

extern "C" int eepacc_nlp_trial(int device, int B, int rows_per_route, const double* r_dev, const double* t_dev, const double* dt_dev,
                                const double* r_trial_dev, const double* alpha_dev, const double* tau_dev, double* t_trial_dev,
                                double* out_dev, void* stream) {
    if (B < 1 || rows_per_route < 1 || !r_dev || !t_dev || !dt_dev || !r_trial_dev || !alpha_dev || !tau_dev || !t_trial_dev || !out_dev)
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_trial: null argument");
    NLPCHK(hipSetDevice(device));
    hipLaunchKernelGGL(k_nlp_trial, dim3(B), dim3(RED_T), 0, (hipStream_t)stream, rows_per_route, r_dev, t_dev, dt_dev, r_trial_dev, alpha_dev,
                       tau_dev, t_trial_dev, out_dev, nullptr, 0);
    NLPCHK(hipGetLastError());
    return EEPACC_OK;
}

#include "eepacc_nlp_solve.inc"
