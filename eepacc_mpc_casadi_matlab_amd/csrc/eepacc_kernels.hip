// eepacc_kernels.hip -- hand-written HIP (gfx950) kernels of the batched ABMPC step.
//
// One QP instance per 64-lane wavefront: horizon stage k lives on lane k (terminal stage on
// lane N), the small inverse-KKT block of the working set and all per-stage vectors are staged
// in LDS.  The horizon condensing is never materialised: every constraint row of the reference
// QP (ABO/Functions/MPCs/CreateQP_AB.m:256-387) is  al*s_k + be*v_k + ga*a_k + de*a_{k-1} - xi <= b
// and the condensed double integrator (ABO/RunOpt_ABMPC.m:74-82 +
// ABO/Functions/MPCs/TransformToDenseFormulation.m:46-68) turns into wave-level prefix /
// suffix scans.  The slack columns are eliminated exactly (capped-multiplier groups, compliant
// row for the quadratic slack); the dense active-set solve is a dual (Goldfarb-Idnani type)
// method on the N x N acceleration block.  DESIGN.md has the derivation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "eepacc_device.h"
#include "eepacc_stage.h"
#include "../../include/eepacc.h"

namespace eepacc {

#ifdef EEPACC_AB_TIMING
__device__ unsigned long long g_ab_prof[16];
#define PTIC(L) long long _pt = wall_clock64()
#define PTOC(L, slot) do { long long _n = wall_clock64(); (L).prof[slot] += _n - _pt; _pt = _n; } while (0)
#else
#define PTIC(L)
#define PTOC(L, slot)
#endif
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

constexpr double kInf = 1e300;
constexpr double kTolViol = 1e-11;
constexpr double kTolDual = 1e-12;
constexpr int kSinglePasses = 8;
constexpr int kChunkStepsDefault = 16;      // MPC steps per work unit of the closed-loop kernel

// ----------------------------------------------------------------------------------------------
// wave primitives
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// broadcast from a wave-uniform source lane (v_readlane)
__device__ __forceinline__ double bcast(double x, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    int lo = __builtin_amdgcn_readlane(__double2loint(x), s);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), s);
    return __hiloint2double(hi, lo);
}

// DPP cross-lane moves (VALU data path, no LDS round trip).  ctrl: row_shr:n = 0x110+n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shr:1 = 0x138 (gfx9-family encodings).
// dpp_zero: lanes without a valid source (or masked rows) read 0; dpp_keep: they keep their value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_zero(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_keep(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane63(double x) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes (Hillis-Steele inside 16-lane rows, then row broadcasts)
__device__ __forceinline__ double scan_incl(double x) {
    x += dpp_zero<0x111, 0xf>(x);
    x += dpp_zero<0x112, 0xf>(x);
    x += dpp_zero<0x114, 0xf>(x);
    x += dpp_zero<0x118, 0xf>(x);
    x += dpp_zero<0x142, 0xa>(x);
    x += dpp_zero<0x143, 0xc>(x);
    return x;
}
__device__ __forceinline__ double wave_sum(double x) { return read_lane63(scan_incl(x)); }
// exclusive prefix sum over lanes
__device__ __forceinline__ double scan_excl(double x) { return dpp_zero<0x138, 0xf>(scan_incl(x)); }
__device__ __forceinline__ double wave_max(double x) {
    x = fmax(x, dpp_keep<0x111, 0xf>(x));
    x = fmax(x, dpp_keep<0x112, 0xf>(x));
    x = fmax(x, dpp_keep<0x114, 0xf>(x));
    x = fmax(x, dpp_keep<0x118, 0xf>(x));
    x = fmax(x, dpp_keep<0x142, 0xa>(x));
    x = fmax(x, dpp_keep<0x143, 0xc>(x));
    return read_lane63(x);
}
// arg-min / arg-max with integer payload; ties go to the lowest lane (deterministic)
__device__ __forceinline__ void wave_argmax(double& v, int& p) {
    const double best = wave_max(v);
    const unsigned long long mask = __ballot(v == best);
    const int src = mask ? (__ffsll((long long)mask) - 1) : 0;
    p = __builtin_amdgcn_readlane(p, src);
    v = best;
}
__device__ __forceinline__ void wave_argmin(double& v, int& p) {
    double nv = -v;
    wave_argmax(nv, p);
    v = -nv;
}

// ----------------------------------------------------------------------------------------------
// row catalogue
__device__ __forceinline__ int group_of(int t) {
    return t < R_AMAX ? G_NONE : (t < R_SAFE1 ? G_F : (t < R_VINC ? G_S : (t == R_VINC ? G_V : G_H)));
}
__device__ __forceinline__ double row_al(int t) {
    return (t == R_SLO) ? -1.0 : ((t == R_SHI || t == R_SAFE1 || t == R_SAFE2 || t == R_HWP) ? 1.0 : 0.0);
}
__device__ __forceinline__ double row_be(int t, double tau_min, double chw) {
    switch (t) {
        case R_VLO: case R_VINC: return -1.0;
        case R_VHI: case R_VLIM: case R_VCURV: case R_VSTOP: case R_VTL: return 1.0;
        case R_SAFE2: return tau_min;
        case R_HWP: return chw;
        default: return 0.0;
    }
}
__device__ __forceinline__ double row_ga(int t) {
    return (t == R_AMAX || t == R_JMAX) ? 1.0 : ((t == R_AMIN || t == R_JMIN) ? -1.0 : 0.0);
}
__device__ __forceinline__ double row_de(int t, int k) {
    if (k == 0) return 0.0;
    return t == R_JMAX ? -1.0 : (t == R_JMIN ? 1.0 : 0.0);
}

template <int MMAX, int NS>
struct WaveMem {                 // one per wave, in LDS (followed by the wave's NS x NS matrix He)
    double P[MMAX * (MMAX + 1) / 2];
    double yv[NS], av[NS];
    double shv[NS + 1], vhv[NS + 1];
    double ub[NS + 1], sub[NS + 1], vub[NS + 1];     // images of a vector; also scratch of adjoint()
    double ws[NS + 1], wv[NS + 1], wa[NS + 1];
    double e_al[MMAX], e_be[MMAX], e_ga[MMAX], e_de[MMAX], e_d[MMAX];
    double lam[MMAX], sv[MMAX], rv[MMAX], colk[MMAX];
    int w_k[MMAX];
};

__device__ __forceinline__ int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// per-lane (= per-stage) registers of the wave's QP
struct Lane {
    int lane, N;
    double T, tau, tau1;          // T_k, tau_k, tau_{k+1}
    double tau_rev;               // tau_{N-lane} (reversed stage order, see adjoint())
    double ba[kNumRowTypes];      // a-space right-hand sides
    unsigned valid;               // bit t: row (t, lane) exists with a non-zero normal
#ifdef EEPACC_AB_TIMING
    long long prof[14];
#endif
    unsigned ign;                 // bit t / 16+g: duplicate row ignored during this solve
    unsigned long long kmask;     // wave-uniform: stages whose penalty q n n' is folded into He
    double lbF, lbS, lbV, lbH;    // slack lower bounds (constant rows of stage 0 fold in here)
    unsigned long long code;      // 4 bits per type: 0 off, 1 in working set, 2 group pivot, 3 compliant
    double chw;                   // headway-policy coefficient T_hwp + G_hwp*v_est(k)
    double g0;                    // base gradient of the condensed objective
    double a, sh, vh, am1;        // acceleration and homogeneous trajectories, a_{k-1}
    int base;                     // first working-set position of this lane's rows
};

__device__ __forceinline__ int code_of(const Lane& L, int t) { return (int)((L.code >> (4 * t)) & 15ull); }
__device__ __forceinline__ void set_code(Lane& L, int t, int c) {
    L.code = (L.code & ~(15ull << (4 * t))) | ((unsigned long long)c << (4 * t));
}

struct Cfg {   // wave-uniform scalars pulled out of DevCfg once
    int N;
    double tau_min, wF, wS, wV, wH, qH;
};

__device__ __forceinline__ double group_w(const Cfg& c, int g) {
    return g == G_F ? c.wF : (g == G_S ? c.wS : (g == G_V ? c.wV : c.wH));
}
__device__ __forceinline__ double group_lb(const Lane& L, int g) {
    return g == G_F ? L.lbF : (g == G_S ? L.lbS : (g == G_V ? L.lbV : L.lbH));
}
__device__ __forceinline__ int lane_group(const Lane& L, int t) { return L.lane == L.N ? G_NONE : group_of(t); }

// value of row t at this lane for the current homogeneous trajectory (without slack)
__device__ __forceinline__ double row_val(const Lane& L, const Cfg& c, int t, double ba_t) {
    return row_al(t) * L.sh + row_be(t, c.tau_min, L.chw) * L.vh + row_ga(t) * L.a + row_de(t, L.lane) * L.am1 - ba_t;
}

// pivot type of linear group g at this lane (-1 if the group is in Z state)
__device__ __forceinline__ int pivot_of(const Lane& L, int g) {
    int p = -1;
#pragma unroll
    for (int t = R_AMAX; t <= R_VINC; ++t)
        if (group_of(t) == g && code_of(L, t) == 2) p = t;
    return p;
}

// homogeneous double-integrator response to the per-lane input x (lane k < N holds x_k):
//   vh_k = sum_{i<k} T_i x_i ,  sh_k = sum_{i<k} (T_i vh_i + T_i^2/2 x_i)
__device__ __forceinline__ void hom_traj(const Lane& L, double x, double& sh, double& vh) {
    double xi = (L.lane < L.N) ? x : 0.0;
    vh = scan_excl(L.T * xi);
    double inc = (L.lane < L.N) ? (L.T * vh + 0.5 * L.T * L.T * xi) : 0.0;
    sh = scan_excl(inc);
}

// out_k = sum_i Hinv[i][k] * yv[i]   (Hinv symmetric, table in LDS, yv in LDS)
template <int NS>
__device__ __forceinline__ double hinv_mul(const double* Hs, const double* yv, int N, int lane) {
    // He is stored NS x NS (zero padded) so that every load has an immediate offset
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double* col = Hs + (lane & (NS - 1));
#pragma unroll
    for (int i = 0; i < NS; i += 4) {
        a0 = fma(col[(i + 0) * NS], yv[i + 0], a0);
        a1 = fma(col[(i + 1) * NS], yv[i + 1], a1);
        a2 = fma(col[(i + 2) * NS], yv[i + 2], a2);
        a3 = fma(col[(i + 3) * NS], yv[i + 3], a3);
    }
    return lane < N ? (a0 + a1) + (a2 + a3) : 0.0;
}

// two products with one pass over the table
template <int NS>
__device__ __forceinline__ void hinv_mul2(const double* Hs, const double* y0, const double* y1, int N, int lane,
                                          double& o0, double& o1) {
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    const double* col = Hs + (lane & (NS - 1));
#pragma unroll
    for (int i = 0; i < NS; i += 2) {
        const double h0 = col[(i + 0) * NS], h1 = col[(i + 1) * NS];
        a0 = fma(h0, y0[i + 0], a0); b0 = fma(h0, y1[i + 0], b0);
        a1 = fma(h1, y0[i + 1], a1); b1 = fma(h1, y1[i + 1], b1);
    }
    o0 = lane < N ? a0 + a1 : 0.0;
    o1 = lane < N ? b0 + b1 : 0.0;
}

// four products with one pass over the table
template <int NS>
__device__ __forceinline__ void hinv_mul4(const double* Hs, const double* y0, const double* y1, const double* y2,
                                          const double* y3, int N, int lane, double (&o)[4]) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    const double* col = Hs + (lane & (NS - 1));
#pragma unroll
    for (int i = 0; i < NS; i += 2) {
        if ((i & 7) == 0) __builtin_amdgcn_sched_barrier(0);      // bound the number of loads in flight (registers)
        const double h0 = col[(i + 0) * NS], h1 = col[(i + 1) * NS];
        a0 = fma(h0, y0[i], a0); a1 = fma(h0, y1[i], a1); a2 = fma(h0, y2[i], a2); a3 = fma(h0, y3[i], a3);
        b0 = fma(h1, y0[i + 1], b0); b1 = fma(h1, y1[i + 1], b1); b2 = fma(h1, y2[i + 1], b2); b3 = fma(h1, y3[i + 1], b3);
    }
    const bool in = lane < N;
    o[0] = in ? a0 + b0 : 0.0; o[1] = in ? a1 + b1 : 0.0; o[2] = in ? a2 + b2 : 0.0; o[3] = in ? a3 + b3 : 0.0;
}

// a-space normal of the row (kq; al,be,ga,de) evaluated at this lane j:
__device__ __forceinline__ double normal_at(const Lane& L, int kq, double al, double be, double ga, double de, double tau_kq) {
    double c = 0.0;
    const int j = L.lane;
    if (j < L.N) {
        if (j < kq) c = L.T * (be + al * (0.5 * L.T + tau_kq - L.tau1));
        if (j == kq) c += ga;
        if (j == kq - 1) c += de;
    }
    return c;
}

// adjoint of the condensing: given stage weights on (s_k, v_k, a_k) in LDS (ws, wv, wa, k = 0..N)
// returns d/da_j of sum_k ws_k s_k + wv_k v_k + wa_k a_k  for lane j < N
template <int NS>
__device__ __forceinline__ double adjoint(const Lane& L, const double* ws, const double* wv, const double* wa, double* tmp) {
    // suffix sums over stages k > j are prefix sums over the reversed stage order: lane r holds
    // stage N - r; the three running sums are written back in stage order through LDS (wa[65..])
    const int r = L.lane, N = L.N;
    const int k = N - r;
    const bool in = k >= 0;
    const double s = in ? ws[k] : 0.0, v = in ? wv[k] : 0.0;
    const double tk = in ? L.tau_rev : 0.0;
    const double WS = scan_excl(s), WV = scan_excl(v), WST = scan_excl(s * tk);
    if (in) { tmp[k] = WS; tmp[(NS + 1) + k] = WV; tmp[2 * (NS + 1) + k] = WST; }
    WSYNC();
    const int j = L.lane;
    double g = 0.0;
    if (j < N) g = wa[j] + L.T * (tmp[(NS + 1) + j] + tmp[2 * (NS + 1) + j] - (L.tau1 - 0.5 * L.T) * tmp[j]);
    WSYNC();
    return g;
}

enum Ev : int { EV_NONE = 0, EV_DROP, EV_COMPL, EV_DROPH, EV_CAP, EV_CAPIN };

struct SolveStats { int status, iters, events, m; };

// The quadratic slack xi_h of stage k, once above its bound, is eliminated into the objective:
// H_eff = H + q * sum_k n_k n_k'  (n_k: a-space normal of the headway-policy row).  He holds
// H_eff^-1 for this wave; adding / removing one stage is a Sherman-Morrison rank-one update.
template <int MMAX, int NS>
__device__ __forceinline__ void he_rank1(const Lane& L, const Cfg& c, WaveMem<MMAX, NS>& M, double* He,
                                         const double* tauv, int k, bool add) {
    const int lane = L.lane, N = L.N;
    const double chwk = bcast(L.chw, k);
    const double nk = normal_at(L, k, 1.0, chwk, 0.0, 0.0, tauv[k]);
    if (lane < NS) M.yv[lane] = nk;
    WSYNC();
    const double y = hinv_mul<NS>(He, M.yv, N, lane);
    double sy, vy;
    hom_traj(L, y, sy, vy);
    const double ny = bcast(sy + chwk * vy, k);                 // n_k' y
    const double kappa = add ? c.qH / (1.0 + c.qH * ny) : -c.qH / (1.0 - c.qH * ny);
    if (lane < NS) M.ub[lane] = y;          // y is zero beyond N
    WSYNC();
    if (lane < NS) {
        const double yj = kappa * y;
        double* col = He + lane;
#pragma unroll
        for (int i = 0; i < NS; i += 4) {
            const double h0 = col[(i + 0) * NS], h1 = col[(i + 1) * NS], h2 = col[(i + 2) * NS], h3 = col[(i + 3) * NS];
            const double y0 = M.ub[i], y1 = M.ub[i + 1], y2 = M.ub[i + 2], y3 = M.ub[i + 3];
            col[(i + 0) * NS] = fma(-y0, yj, h0); col[(i + 1) * NS] = fma(-y1, yj, h1);
            col[(i + 2) * NS] = fma(-y2, yj, h2); col[(i + 3) * NS] = fma(-y3, yj, h3);
        }
    }
    WSYNC();
}

template <int MMAX, int NS>
__device__ __forceinline__ void he_sync(Lane& L, const Cfg& c, WaveMem<MMAX, NS>& M, double* He, const double* tauv) {
    const unsigned long long want = __ballot(L.lane < L.N && code_of(L, R_HWP) == 3);
    unsigned long long diff = want ^ L.kmask;
    while (diff) {
        const int k = __ffsll((long long)diff) - 1;
        diff &= diff - 1;
        he_rank1(L, c, M, He, tauv, k, ((want >> k) & 1ull) != 0ull);
    }
    L.kmask = want;
}

template <int NS>
__device__ __forceinline__ void he_load_base(double* He, const double* __restrict__ base, int N, int lane) {
#pragma unroll 4
    for (int e = lane; e < NS * NS; e += 64) {
        const int i = e / NS, j = e % NS;
        He[e] = (i < N && j < N) ? base[i * N + j] : 0.0;
    }
}

// row/column of entry e of a packed lower triangle (e = r(r+1)/2 + c), one table per workgroup
template <int MMAX>
__device__ __forceinline__ unsigned short* rc_table() {
    __shared__ unsigned short tab[MMAX * (MMAX + 1) / 2];
    return tab;
}
template <int MMAX>
__device__ __forceinline__ void rc_table_init() {      // every thread of the workgroup, before any returns
    unsigned short* tab = rc_table<MMAX>();
    for (int e = threadIdx.x; e < MMAX * (MMAX + 1) / 2; e += blockDim.x) {
        int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (r * (r + 1) / 2 > e) --r;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        tab[e] = (unsigned short)((r << 8) | (e - r * (r + 1) / 2));
    }
    __syncthreads();
}

// ----------------------------------------------------------------------------------------------
// rebuild the working-set list + effective rows from the state codes, build S = C Hinv C' + D,
// invert it in place (symmetric sweeps).  returns m (or -1 if S was numerically singular).
// fast = 1: the last event appended the plain row (kq, tq) whose column sv = C u, rv = P sv and pivot
// zz = c'u - sv'rv are still in LDS -> bordered update of P;  fast = 2: it dropped the row at list
// position drop_pos -> rank-one downdate;  fast = 0 (or any inconsistency): full rebuild.
struct FastInfo { int fast, m_old, kq, tq, drop_pos; double zz; };

template <int MMAX, int NS>
__device__ __forceinline__ int rebuild_and_factor(Lane& L, const Cfg& c, WaveMem<MMAX, NS>& M, double* Hs, const double* tauv,
                                                  const FastInfo& F) {
    const int lane = L.lane, N = L.N;
#ifdef EEPACC_AB_TIMING
    long long _q = wall_clock64();
#define RTOC(slot) do { long long _n = wall_clock64(); L.prof[slot] += _n - _q; _q = _n; } while (0)
#else
#define RTOC(slot)
#endif
    he_sync(L, c, M, Hs, tauv);
    RTOC(9);
    // count this lane's active rows
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < kNumRowTypes; ++t) {
        int cd = code_of(L, t);
        cnt += (cd == 1) ? 1 : 0;
    }
    double basef = scan_excl((double)cnt);
    L.base = (int)(basef + 0.5);
    int m = (int)(wave_sum((double)cnt) + 0.5);
    if (m > MMAX) return -2;
    const int pF = pivot_of(L, G_F), pS = pivot_of(L, G_S), pV = pivot_of(L, G_V);
    int pos = L.base;
#pragma unroll
    for (int t = 0; t < kNumRowTypes; ++t) {
        int cd = code_of(L, t);
        if (cd == 1) {
            int g = lane_group(L, t);
            double al = row_al(t), be = row_be(t, c.tau_min, L.chw), ga = row_ga(t), de = row_de(t, lane);
            double d = L.ba[t];
            if (g == G_H) {
                d += L.lbH;
            } else if (g != G_NONE) {
                int p = (g == G_F) ? pF : (g == G_S ? pS : pV);
                if (p >= 0) {
                    double pba = 0.0;
#pragma unroll
                    for (int u = R_AMAX; u <= R_VINC; ++u) if (u == p) pba = L.ba[u];
                    al -= row_al(p); be -= row_be(p, c.tau_min, L.chw); ga -= row_ga(p); de -= row_de(p, lane);
                    d -= pba;
                } else {
                    d += group_lb(L, g);
                }
            }
            M.e_al[pos] = al; M.e_be[pos] = be; M.e_ga[pos] = ga; M.e_de[pos] = de;
            M.e_d[pos] = d; M.w_k[pos] = lane;
            ++pos;
        }
    }
    WSYNC();
    RTOC(10);
    if (m == 0) return 0;
    const unsigned short* rc = rc_table<MMAX>();
    if (F.fast == 1 && m == F.m_old + 1 && F.m_old > 0) {
        // position of the new row in the new (lane-major) list
        int pl = 0;
#pragma unroll
        for (int t = 0; t < kNumRowTypes; ++t) pl += (t < F.tq && code_of(L, t) == 1) ? 1 : 0;
        const int p = __builtin_amdgcn_readlane(L.base + pl, __builtin_amdgcn_readfirstlane(F.kq));
        const double iz = 1.0 / F.zz;
        const int nnz = m * (m + 1) / 2;
        // in place, highest entries first: an entry moves to a higher packed index, so a chunk never
        // overwrites what a later (lower) chunk still has to read
        for (int e0 = ((nnz - 1) >> 6) << 6; e0 >= 0; e0 -= 64) {
            const int e = e0 + lane;
            double v = 0.0;
            if (e < nnz) {
                const int code = rc[e], r = code >> 8, cc = code & 255;
                const int i = r < p ? r : r - 1, j = cc < p ? cc : cc - 1;
                if (r == p && cc == p) v = iz;
                else if (r == p) v = -M.rv[j] * iz;
                else if (cc == p) v = -M.rv[i] * iz;
                else v = M.P[pidx(i, j)] + M.rv[i] * M.rv[j] * iz;
            }
            WSYNC();
            if (e < nnz) M.P[e] = v;
            WSYNC();
        }
        RTOC(12);
        return m;
    }
    if (F.fast == 2 && m == F.m_old - 1) {
        const int p = F.drop_pos, mo = F.m_old;
        if (lane < mo) M.colk[lane] = M.P[pidx(lane, p)];
        WSYNC();
        const double ip = 1.0 / M.colk[p];
        const int nnz = m * (m + 1) / 2;
        // lowest entries first: an entry moves to a lower packed index
        for (int e0 = 0; e0 < nnz; e0 += 64) {
            const int e = e0 + lane;
            double v = 0.0;
            if (e < nnz) {
                const int code = rc[e], r = code >> 8, cc = code & 255;
                const int i = r < p ? r : r + 1, j = cc < p ? cc : cc + 1;
                v = M.P[pidx(i, j)] - M.colk[i] * M.colk[j] * ip;
            }
            WSYNC();
            if (e < nnz) M.P[e] = v;
            WSYNC();
        }
        RTOC(12);
        return m;
    }
    // S columns: u_j = He c_j, two columns per pass: each He element is loaded once for both products,
    // the scan chains of the two trajectories overlap, and row i picks the images at its stage with
    // lane shuffles (no LDS round trip).  Input vectors in yv | lam (free while the factor is rebuilt).
    {
        const int ki = lane < m ? M.w_k[lane] : 0;
        const int kim1 = ki > 0 ? ki - 1 : 0;
        const double eal = lane < m ? M.e_al[lane] : 0.0, ebe = lane < m ? M.e_be[lane] : 0.0;
        const double ega = (lane < m && ki < N) ? M.e_ga[lane] : 0.0;
        const double ede = (lane < m && ki > 0 && ki <= N) ? M.e_de[lane] : 0.0;
        for (int j = 0; j < m; j += 2) {
            const bool two = j + 1 < m;
            const int j1 = two ? j + 1 : j;
            const int kj0 = M.w_k[j], kj1 = M.w_k[j1];
            const double c0 = normal_at(L, kj0, M.e_al[j], M.e_be[j], M.e_ga[j], M.e_de[j], tauv[kj0]);
            const double c1 = normal_at(L, kj1, M.e_al[j1], M.e_be[j1], M.e_ga[j1], M.e_de[j1], tauv[kj1]);
            if (lane < NS) { M.yv[lane] = c0; M.lam[lane] = c1; }
            WSYNC();
            double u0, u1;
            hinv_mul2<NS>(Hs, M.yv, M.lam, N, lane, u0, u1);
            WSYNC();
            double su0, vu0, su1, vu1;
            hom_traj(L, u0, su0, vu0);
            hom_traj(L, u1, su1, vu1);
            const double sx = eal * __shfl(su0, ki, 64) + ebe * __shfl(vu0, ki, 64) + ega * __shfl(u0, ki, 64) + ede * __shfl(u0, kim1, 64);
            const double sy = eal * __shfl(su1, ki, 64) + ebe * __shfl(vu1, ki, 64) + ega * __shfl(u1, ki, 64) + ede * __shfl(u1, kim1, 64);
            if (lane >= j && lane < m) M.P[pidx(lane, j)] = sx;
            if (two && lane >= j + 1 && lane < m) M.P[pidx(lane, j + 1)] = sy;
        }
        WSYNC();
    }
    RTOC(11);
    // in-place inversion by symmetric sweeps: after sweeping every pivot P = -S^-1
    int singular = 0;
    if (lane < m) M.sv[lane] = fabs(M.P[pidx(lane, lane)]);     // original diagonal (pivot scale)
    WSYNC();
    // the packed lower triangle is spread over all 64 lanes (entry e = lane + 64 t; its row and column
    // come from a small table), so a sweep costs m(m+1)/128 entry updates per lane instead of m
    const int nnz = m * (m + 1) / 2;
    for (int k = 0; k < m; ++k) {
        const double d = M.P[pidx(k, k)];
        if (!(d > 1e-12 * M.sv[k])) { singular = 1; break; }
        const double inv = 1.0 / d;
        if (lane < m) M.colk[lane] = M.P[pidx(lane, k)];
        WSYNC();
#pragma unroll 2
        for (int e = lane; e < nnz; e += 64) {
            const int code = rc[e], r = code >> 8, cc = code & 255;
            const double c0 = M.colk[r];
            const double cl = M.colk[cc] * inv;
            double v0 = M.P[e] - c0 * cl;
            if (cc == k) v0 = c0 * inv;
            if (r == k) v0 = (cc == k) ? -inv : cl;
            M.P[e] = v0;
        }
        WSYNC();
    }
    RTOC(12);
    if (singular) return -1;
    if (lane < m)
        for (int r = lane; r < m; ++r) M.P[pidx(r, lane)] = -M.P[pidx(r, lane)];
    WSYNC();
    return m;
}

// gradient-side vector: g_eff + (incoming multiplier) * c_q + C' lam   evaluated per lane
template <int MMAX, int NS>
__device__ __forceinline__ double gradient_side(const Lane& L, const Cfg& c, WaveMem<MMAX, NS>& M, int m, bool with_lam,
                                double lam_q, int kq, double qal, double qbe, double qga, double qde) {
    const int lane = L.lane, N = L.N;
    // pivot rows act like working-set rows with multiplier w_g (own stage: plain stores)
    double s = 0.0, v = 0.0, a0 = 0.0, a1 = 0.0;   // a1 -> stage lane-1
    if (lane < N) {
#pragma unroll
        for (int t = R_AMAX; t <= R_VINC; ++t) {
            if (code_of(L, t) == 2) {
                double w = group_w(c, group_of(t));
                s += w * row_al(t); v += w * row_be(t, c.tau_min, L.chw);
                a0 += w * row_ga(t); a1 += w * row_de(t, lane);
            }
        }
    }
    if (lane < N && code_of(L, R_HWP) == 3) {
        // quadratic slack above its bound: penalty w*xi + q/2 xi^2, xi = n'a - b; linear part (w - q b) n
        const double wl = c.wH - c.qH * L.ba[R_HWP];
        s += wl * row_al(R_HWP); v += wl * row_be(R_HWP, c.tau_min, L.chw);
    }
    if (lane <= N) { M.ws[lane] = s; M.wv[lane] = v; M.wa[lane] = a0; }
    WSYNC();
    if (lane > 0 && lane < N && a1 != 0.0) atomicAdd(&M.wa[lane - 1], a1);
    if (with_lam && lane < m) {
        const int ki = M.w_k[lane];
        const double l = M.lam[lane];
        atomicAdd(&M.ws[ki], l * M.e_al[lane]);
        atomicAdd(&M.wv[ki], l * M.e_be[lane]);
        if (ki < N && M.e_ga[lane] != 0.0) atomicAdd(&M.wa[ki], l * M.e_ga[lane]);
        if (ki > 0 && M.e_de[lane] != 0.0) atomicAdd(&M.wa[ki - 1], l * M.e_de[lane]);
    }
    if (lam_q != 0.0 && lane == 0) {
        atomicAdd(&M.ws[kq], lam_q * qal);
        atomicAdd(&M.wv[kq], lam_q * qbe);
        if (kq < N && qga != 0.0) atomicAdd(&M.wa[kq], lam_q * qga);
        if (kq > 0 && qde != 0.0) atomicAdd(&M.wa[kq - 1], lam_q * qde);
    }
    WSYNC();
    double g = adjoint<NS>(L, M.ws, M.wv, M.wa, M.ub);
    return (lane < N) ? g + L.g0 : 0.0;
}

// C x for the working-set rows (x given through LDS images x / sx / vx): result for lane i < m
template <int MMAX, int NS>
__device__ __forceinline__ double rows_dot_img(const WaveMem<MMAX, NS>& M, int i, int N, const double* x,
                                               const double* sx, const double* vx) {
    const int ki = M.w_k[i];
    double s = M.e_al[i] * sx[ki] + M.e_be[i] * vx[ki];
    if (ki < N) s += M.e_ga[i] * x[ki];
    if (ki > 0) s += M.e_de[i] * x[ki - 1];
    return s;
}
template <int MMAX, int NS>
__device__ __forceinline__ double rows_dot(const WaveMem<MMAX, NS>& M, int i, int N) {
    return rows_dot_img(M, i, N, M.ub, M.sub, M.vub);
}

// lam = -P (d + C h (+ nothing else)); h given through ub/sub/vub
template <int MMAX, int NS>
__device__ __forceinline__ void solve_multipliers(WaveMem<MMAX, NS>& M, int m, int lane, int N) {
    if (lane < m) M.sv[lane] = M.e_d[lane] + rows_dot(M, lane, N);
    WSYNC();
    if (lane < m) {
        double acc = 0.0;
        for (int j = 0; j < m; ++j) acc = fma(M.P[pidx(lane, j)], M.sv[j], acc);
        M.lam[lane] = -acc;
    }
    WSYNC();
}

// primal point from the multipliers: a = -Hinv (g_eff + lam_q c_q + C' lam); also refreshes the
// homogeneous trajectories and their LDS images (av/shv/vhv)
template <int MMAX, int NS>
__device__ __forceinline__ void primal_from_multipliers(Lane& L, const Cfg& c, WaveMem<MMAX, NS>& M, const double* Hs, int m,
                                        double lam_q, int kq, double qal, double qbe, double qga, double qde,
                                        double& grad_total) {
    double g = gradient_side(L, c, M, m, true, lam_q, kq, qal, qbe, qga, qde);
    grad_total = g;
    if (L.lane < NS) M.yv[L.lane] = g;
    WSYNC();
    L.a = -hinv_mul<NS>(Hs, M.yv, L.N, L.lane);
    hom_traj(L, L.a, L.sh, L.vh);
    L.am1 = dpp_zero<0x138, 0xf>(L.a);
    if (L.lane < L.N) M.av[L.lane] = L.a;
    if (L.lane <= L.N) { M.shv[L.lane] = L.sh; M.vhv[L.lane] = L.vh; }
    WSYNC();
}

// primal point + iterative refinement: the multipliers come out of the explicit inverse P, whose
// accuracy degrades with cond(S) (many active rows, stiff ORIG weights); the residual of the
// working-set equations  C a - D lam = d  is evaluated exactly from the scans and fed back
// through P until it is at rounding level.  Stationarity holds by construction of a.
template <int MMAX, int NS>
__device__ __forceinline__ void refine_primal(Lane& L, const Cfg& c, WaveMem<MMAX, NS>& M, const double* Hs, int m,
                                              double lam_q, int kq, double qal, double qbe, double qga, double qde,
                                              double& grad_total, int max_rounds, double res_tol) {
    primal_from_multipliers(L, c, M, Hs, m, lam_q, kq, qal, qbe, qga, qde, grad_total);
    if (m == 0) return;
    for (int round = 0; round < max_rounds; ++round) {
        double res = 0.0, rel = 0.0;
        if (L.lane < m) {
            res = rows_dot_img(M, L.lane, L.N, M.av, M.shv, M.vhv) - M.e_d[L.lane];
            rel = fabs(res) / (1.0 + fabs(M.e_d[L.lane]));
            M.sv[L.lane] = res;
        }
        int dummy = L.lane;
        wave_argmax(rel, dummy);
        if (!(rel > res_tol)) break;
        WSYNC();
        if (L.lane < m) {
            double acc = 0.0;
            for (int j = 0; j < m; ++j) acc = fma(M.P[pidx(L.lane, j)], M.sv[j], acc);
            M.lam[L.lane] += acc;
        }
        WSYNC();
        primal_from_multipliers(L, c, M, Hs, m, lam_q, kq, qal, qbe, qga, qde, grad_total);
    }
}

// slack of linear group g at this lane for the current point
__device__ __forceinline__ double group_xi(const Lane& L, const Cfg& c, int g) {
    int p = pivot_of(L, g);
    if (p < 0) return group_lb(L, g);
    double pba = 0.0;
#pragma unroll
    for (int u = R_AMAX; u <= R_VINC; ++u) if (u == p) pba = L.ba[u];
    return row_val(L, c, p, pba);
}

// ----------------------------------------------------------------------------------------------
// the dual active-set solve.  On entry L.code holds the (warm) working set; on exit the optimal
// one, L.a the accelerations, L.lamt the multipliers per (lane, type).
// repair of dual infeasibilities of a warm working set (multipliers in L.lamt).  The first passes
// fix only the worst one (a single wrong row usually drags many multipliers negative; dropping
// them all would throw the warm start away), later passes fix all of them at once.
template <int MMAX, int NS>
__device__ __forceinline__ int warm_repair(Lane& L, const Cfg& c, const WaveMem<MMAX, NS>& M, int m, bool single) {
    const int lane = L.lane, N = L.N;
    double lmax = 0.0;
    if (lane < m) lmax = fabs(M.lam[lane]);
    lmax = wave_max(lmax);
    const double tol = kTolDual * (1.0 + lmax);
    int changed = 0;
    double sumF = 0.0, sumS = 0.0, sumV = 0.0;
    double bestF = -1e300, bestS = -1e300, bestV = -1e300;     // largest multiplier per group
    int bF = -1, bS = -1, bV = -1;
    double worst = tol; int fix = 0x7fffffff;
    int pos = L.base;
#pragma unroll
    for (int t = 0; t < kNumRowTypes; ++t) {
        int cd = code_of(L, t);
        if (cd != 1) continue;
        int g2 = lane_group(L, t);
        double l = M.lam[pos++];
        if (g2 == G_H) {
            if (-l > tol) {
                if (!single) { set_code(L, t, 0); changed = 1; } else if (-l > worst) { worst = -l; fix = (EV_DROP << 16) | (lane << 5) | t; }
            } else if (l - c.wH > tol) {
                if (!single) { set_code(L, t, 3); changed = 1; } else if (l - c.wH > worst) { worst = l - c.wH; fix = (EV_COMPL << 16) | (lane << 5) | t; }
            }
        } else {
            if (-l > tol) {
                if (!single) { set_code(L, t, 0); changed = 1; l = 0.0; } else if (-l > worst) { worst = -l; fix = (EV_DROP << 16) | (lane << 5) | t; }
            } else if (g2 == G_F) { if (l > bestF) { bestF = l; bF = t; } }
            else if (g2 == G_S) { if (l > bestS) { bestS = l; bS = t; } }
            else if (g2 == G_V) { if (l > bestV) { bestV = l; bV = t; } }
            if (g2 == G_F) sumF += l; else if (g2 == G_S) sumS += l; else if (g2 == G_V) sumV += l;
        }
    }
    // group margins (bound multiplier in Z, pivot multiplier in P)
    int capg = 0;
    if (lane < N) {
#pragma unroll
        for (int g2 = G_F; g2 <= G_V; ++g2) {
            double sum = g2 == G_F ? sumF : (g2 == G_S ? sumS : sumV);
            double w = group_w(c, g2);
            double viol = sum - w;
            if (viol > tol * (1.0 + w)) {
                if (single) { if (viol > worst) { worst = viol; fix = (EV_CAP << 16) | (lane << 5) | g2; } }
                else if (!changed && !capg) capg = g2;
            }
        }
    }
    int el = -1, et = capg;
    if (single) {
        wave_argmax(worst, fix);
        if (fix != 0x7fffffff) {
            changed = 1;
            const int ek = fix >> 16;
            el = (fix >> 5) & 63; et = fix & 31;
            if (lane == el) {
                if (ek == EV_DROP) set_code(L, et, 0);
                else if (ek == EV_COMPL) set_code(L, et, 3);
            }
            if (ek != EV_CAP) el = -1;
        } else el = -1;
    } else if (capg) { el = lane; changed = 1; }
    if (lane == el) {          // group cap violated: make the member with the largest multiplier the pivot
        int p = pivot_of(L, et);
        if (p >= 0) set_code(L, p, 0);
        int bestt = et == G_F ? bF : (et == G_S ? bS : bV);
        if (bestt >= 0 && code_of(L, bestt) == 1) set_code(L, bestt, 2);
    }
    return __any(changed);
}

// most violated inactive row / slack bound: returns lane*32 + code (code: row type, or 16+group for
// the bound of a slack) or -1; best = its scaled violation
__device__ __forceinline__ int find_violation(const Lane& L, const Cfg& c, double tolv, double& best) {
    const int lane = L.lane, N = L.N;
    double xiF = 0, xiS = 0, xiV = 0;
    if (lane < N) { xiF = group_xi(L, c, G_F); xiS = group_xi(L, c, G_S); xiV = group_xi(L, c, G_V); }
    double myb = tolv; int myp = -1;
#pragma unroll
    for (int t = 0; t < kNumRowTypes; ++t) {
        if (!((L.valid >> t) & 1u) || ((L.ign >> t) & 1u)) continue;
        if (code_of(L, t) != 0) continue;
        int g2 = lane_group(L, t);
        double val = row_val(L, c, t, L.ba[t]);
        val -= (g2 == G_F) ? xiF : (g2 == G_S ? xiS : (g2 == G_V ? xiV : (g2 == G_H ? L.lbH : 0.0)));
        double sc = val / (1.0 + fabs(L.ba[t]));
        if (sc > myb) { myb = sc; myp = t; }
    }
    if (lane < N) {
        if (pivot_of(L, G_F) >= 0 && !((L.ign >> (16 + G_F)) & 1u) && L.lbF - xiF > myb) { myb = L.lbF - xiF; myp = 16 + G_F; }
        if (pivot_of(L, G_S) >= 0 && !((L.ign >> (16 + G_S)) & 1u) && L.lbS - xiS > myb) { myb = L.lbS - xiS; myp = 16 + G_S; }
        if (pivot_of(L, G_V) >= 0 && !((L.ign >> (16 + G_V)) & 1u) && L.lbV - xiV > myb) { myb = L.lbV - xiV; myp = 16 + G_V; }
        // penalised quadratic slack: xi_h = n'a - b must stay above its bound
        if (code_of(L, R_HWP) == 3 && !((L.ign >> (16 + G_H)) & 1u)) {
            double xih = row_val(L, c, R_HWP, L.ba[R_HWP]);
            if (L.lbH - xih > myb) { myb = L.lbH - xih; myp = 16 + G_H; }
        }
    }
    best = myb;
    int bp = (myp < 0) ? 0x7fffffff : (lane * 32 + myp);
    wave_argmax(best, bp);
    return bp == 0x7fffffff ? -1 : bp;
}

struct Incoming { int kq, qcode, tq, gq; bool is_bound; double al, be, ga, de, d; };

// effective a-space row of the incoming constraint for the current group states (lane kq computes,
// everyone receives)
__device__ __forceinline__ void incoming_row(const Lane& L, const Cfg& c, Incoming& q) {
    double qal = 0, qbe = 0, qga = 0, qde = 0, qd = 0;
    const int lane = L.lane;
    if (lane == q.kq) {
        if (!q.is_bound) {
            const int tq = q.tq;
            qal = row_al(tq); qbe = row_be(tq, c.tau_min, L.chw); qga = row_ga(tq); qde = row_de(tq, lane);
            double bq = 0.0;
#pragma unroll
            for (int u = 0; u < kNumRowTypes; ++u) if (u == tq) bq = L.ba[u];
            qd = bq;
            if (q.gq == G_H) {
                qd += L.lbH;
            } else if (q.gq != G_NONE) {
                int p = pivot_of(L, q.gq);
                if (p >= 0) {
                    double pba = 0.0;
#pragma unroll
                    for (int u = R_AMAX; u <= R_VINC; ++u) if (u == p) pba = L.ba[u];
                    qal -= row_al(p); qbe -= row_be(p, c.tau_min, L.chw); qga -= row_ga(p); qde -= row_de(p, lane);
                    qd -= pba;
                } else qd += group_lb(L, q.gq);
            }
        } else if (q.gq == G_H) {
            qal = -row_al(R_HWP); qbe = -row_be(R_HWP, c.tau_min, L.chw);
            qd = -(L.ba[R_HWP] + L.lbH);
        } else {
            int p = pivot_of(L, q.gq);
            double pba = 0.0;
#pragma unroll
            for (int u = R_AMAX; u <= R_VINC; ++u) if (u == p) pba = L.ba[u];
            qal = -row_al(p); qbe = -row_be(p, c.tau_min, L.chw); qga = -row_ga(p); qde = -row_de(p, lane);
            qd = -(pba + group_lb(L, q.gq));
        }
    }
    q.al = bcast(qal, q.kq); q.be = bcast(qbe, q.kq); q.ga = bcast(qga, q.kq); q.de = bcast(qde, q.kq);
    q.d = bcast(qd, q.kq);
}

// ----------------------------------------------------------------------------------------------
// the dual active-set solve.  On entry L.code holds the (warm) working set; on exit the optimal
// one, L.a the accelerations, L.lamt the multipliers per (lane, type).  One loop, one working-set
// change per pass: [rebuild + factor] -> [multipliers] -> [primal + refinement] -> either repair the
// warm start, or pick the next violated constraint / continue the current one and take the step.
template <int MMAX, int NS>
__device__ __forceinline__ SolveStats solve_qp(Lane& L, const Cfg& c, WaveMem<MMAX, NS>& M, double* Hs, const double* Hbase,
                               const double* tauv, int max_iter, double& grad_total) {
    const int lane = L.lane, N = L.N;
    SolveStats st{0, 0, 0, 0};
    int m = 0;
    bool warm = true, have_q = false;
    int pass = 0;
    double lam_q = 0.0, best = 0.0;
    Incoming q{0, 0, 0, 0, false, 0, 0, 0, 0, 0};
    FastInfo F{0, 0, 0, 0, 0, 1.0};   // what the last event did to the working set (see rebuild_and_factor)
    int fast_run = 0;
    for (;;) {
        PTIC(L);
#ifdef EEPACC_AB_TIMING
        L.prof[13] += 1; if (F.fast == 1) L.prof[3] += 1000000; if (F.fast == 2) L.prof[4] += 1000000;
#endif
        // the incremental updates are exact in exact arithmetic; a full rebuild every few of them keeps
        // rounding from accumulating (the refinement below absorbs what is left)
        if (F.fast != 0 && ++fast_run > 6) F.fast = 0;
        if (F.fast == 0) fast_run = 0;
        F.m_old = m;
        m = rebuild_and_factor(L, c, M, Hs, tauv, F);
        F.fast = 0;
        PTOC(L, 0);
        if (m < 0) {
            if (!warm) { st.status = 2; break; }
            L.code = 0ull;                 // unusable warm start: cold start
            he_load_base<NS>(Hs, Hbase, N, lane);
            L.kmask = 0ull;
            WSYNC();
            warm = false;
            continue;
        }
        if (have_q) incoming_row(L, c, q);
        if (m > 0) {
            // multipliers of the working set for the current incoming multiplier
            double g = gradient_side(L, c, M, 0, false, lam_q, q.kq, q.al, q.be, q.ga, q.de);
            if (lane < NS) M.yv[lane] = g;
            WSYNC();
            double h = hinv_mul<NS>(Hs, M.yv, N, lane);
            double shh, vhh;
            hom_traj(L, h, shh, vhh);
            if (lane < N) M.ub[lane] = h;
            if (lane <= N) { M.sub[lane] = shh; M.vub[lane] = vhh; }
            WSYNC();
            solve_multipliers(M, m, lane, N);
        }
        PTOC(L, 1);
        // working accuracy while the working set is still changing; polished after convergence
        refine_primal(L, c, M, Hs, m, lam_q, q.kq, q.al, q.be, q.ga, q.de, grad_total, 3, 1e-11);
        PTOC(L, 2);
        if (warm) {
            if (m > 0) {
                const bool rep = warm_repair(L, c, M, m, pass < kSinglePasses);
                PTOC(L, 3);
                if (rep) {
                    if (++pass >= kSinglePasses + 6) {
                        L.code = 0ull;
                        he_load_base<NS>(Hs, Hbase, N, lane);
                        L.kmask = 0ull;
                        WSYNC();
                        warm = false;
                    }
                    continue;
                }
            }
            warm = false;
        }
        if (!have_q) {
            // Anti-cycling: rounding noise of the order of (largest multiplier) x eps can flip rows in
            // and out at the tightest tolerance (seen with the ORIG weights, w_f = 1e7); the tolerance
            // is relaxed decade by decade if the iteration count shows that this is happening (never
            // beyond 1e-8, scaled by 1+|b|).
            const int relax_every = 3 * N + 30;
            const double tolv = kTolViol * (st.iters < relax_every ? 1.0 : (st.iters < 2 * relax_every ? 10.0 : (st.iters < 3 * relax_every ? 100.0 : 1000.0)));
            const int bp = find_violation(L, c, tolv, best);
            PTOC(L, 4);
            if (bp < 0) break;
            if (++st.iters > max_iter) { st.status = 2; break; }
            q.kq = bp >> 5; q.qcode = bp & 31;
            q.is_bound = q.qcode >= 16;
            q.tq = q.is_bound ? 0 : q.qcode;
            q.gq = q.is_bound ? (q.qcode - 16) : ((q.kq == N) ? G_NONE : group_of(q.qcode));
            have_q = true; lam_q = 0.0;
            incoming_row(L, c, q);
        }
        if (++st.events > 40 * max_iter) { st.status = 2; break; }
        const int kq = q.kq;
        double viol = q.al * M.shv[kq] + q.be * M.vhv[kq] - q.d;
        if (kq < N) viol += q.ga * M.av[kq];
        if (kq > 0) viol += q.de * M.av[kq - 1];
        // u = He c_q and its trajectories
        double cj = normal_at(L, kq, q.al, q.be, q.ga, q.de, tauv[kq]);
        if (lane < NS) M.yv[lane] = cj;
        WSYNC();
        double u = hinv_mul<NS>(Hs, M.yv, N, lane);
        double su, vu;
        hom_traj(L, u, su, vu);
        if (lane < N) M.ub[lane] = u;
        if (lane <= N) { M.sub[lane] = su; M.vub[lane] = vu; }
        WSYNC();
        double cu = q.al * M.sub[kq] + q.be * M.vub[kq];
        if (kq < N) cu += q.ga * M.ub[kq];
        if (kq > 0) cu += q.de * M.ub[kq - 1];
        double sr = 0.0;
        if (m > 0) {
            if (lane < m) M.sv[lane] = rows_dot(M, lane, N);
            WSYNC();
            double r = 0.0;
            if (lane < m) {
                for (int j = 0; j < m; ++j) r = fma(M.P[pidx(lane, j)], M.sv[j], r);
                M.rv[lane] = r;
                sr = M.sv[lane] * r;
            }
            WSYNC();
            sr = wave_sum(sr);
        }
        const double zz = cu - sr;
        double t2 = (zz > 1e-8 * cu) ? viol / zz : kInf;
        if (viol <= 0.0) t2 = 0.0;
        // blocking events, evaluated per (lane, type); multipliers and their rates are read from the
        // working-set list in this lane's order
        double t1 = kInf; int ev = 0x7fffffff;
        {
            double sumLF = 0, sumLS = 0, sumLV = 0, sumRF = 0, sumRS = 0, sumRV = 0;
            int pos = L.base;
#pragma unroll
            for (int t = 0; t < kNumRowTypes; ++t) {
                int cd = code_of(L, t);
                if (cd != 1) continue;
                int g2 = lane_group(L, t);
                const double l = M.lam[pos], r = M.rv[pos];
                ++pos;
                if (g2 == G_H) {
                    // rigid row of the quadratic slack: 0 <= lambda <= w
                    if (r > 0.0) { double tt = fmax(l, 0.0) / r; if (tt < t1) { t1 = tt; ev = (EV_DROP << 16) | (lane << 5) | t; } }
                    else if (r < 0.0) { double tt = fmax(c.wH - l, 0.0) / (-r); if (tt < t1) { t1 = tt; ev = (EV_COMPL << 16) | (lane << 5) | t; } }
                } else {
                    if (r > 0.0) { double tt = fmax(l, 0.0) / r; if (tt < t1) { t1 = tt; ev = (EV_DROP << 16) | (lane << 5) | t; } }
                    if (g2 == G_F) { sumLF += l; sumRF += r; } else if (g2 == G_S) { sumLS += l; sumRS += r; } else if (g2 == G_V) { sumLV += l; sumRV += r; }
                }
            }
            if (lane < N) {
#pragma unroll
                for (int g2 = G_F; g2 <= G_V; ++g2) {
                    double sl = g2 == G_F ? sumLF : (g2 == G_S ? sumLS : sumLV);
                    double srr = g2 == G_F ? sumRF : (g2 == G_S ? sumRS : sumRV);
                    double rate = -srr, margin = group_w(c, g2) - sl;
                    if (lane == kq && q.gq == g2) { rate += 1.0; margin -= lam_q; }
                    if (rate > 0.0) {
                        double tt = fmax(margin, 0.0) / rate;
                        if (tt < t1) { t1 = tt; ev = (EV_CAP << 16) | (lane << 5) | g2; }
                    }
                }
            }
            if (lane == kq && !q.is_bound && q.gq == G_H) {
                double tt = fmax(c.wH - lam_q, 0.0);
                if (tt < t1) { t1 = tt; ev = (EV_CAPIN << 16) | (lane << 5); }
            }
            if (lane == kq && q.is_bound && q.gq == G_H) {
                // incoming slack bound of a penalised row: the row's own multiplier
                // w + q*xi - mu must stay >= 0 while xi rises with the step
                const double xi_now = L.lbH - viol, den = 1.0 - c.qH * zz;
                if (den > 0.0) {
                    double tt = fmax(c.wH + c.qH * xi_now - lam_q, 0.0) / den;
                    if (tt < t1) { t1 = tt; ev = (EV_DROPH << 16) | (lane << 5) | R_HWP; }
                }
            }
            wave_argmin(t1, ev);
        }
        const double tstep = fmin(t1, t2);
        if (!(tstep < 1e299)) {
            // the incoming normal lies in the span of the working set and nothing can be dropped.
            // With a real violation the QP is infeasible; with a rounding-level one the row is a
            // duplicate of active rows (e.g. a_k pinned by an acceleration AND a jerk limit): mark it
            // as ignored for this solve.
            if (best < 1e-7) { if (lane == kq) L.ign |= (1u << q.qcode); have_q = false; lam_q = 0.0; continue; }
            st.status = 1; break;
        }
        lam_q += tstep;
        bool finished = false;
        if (t2 <= t1) {
            // full step: the incoming constraint becomes active
            if (!q.is_bound && m > 0) { F.fast = 1; F.kq = kq; F.tq = q.tq; F.zz = zz; }
            if (lane == kq) {
                if (!q.is_bound) set_code(L, q.tq, 1);
                else if (q.gq == G_H) set_code(L, R_HWP, 1);      // penalised row turns rigid
                else { int p = pivot_of(L, q.gq); set_code(L, p, 1); }
            }
            finished = true;
        } else {
            const int ek = ev >> 16, el = (ev >> 5) & 63, et = ev & 31;
            if (ek == EV_DROP) {
                int pl = 0;
#pragma unroll
                for (int t = 0; t < kNumRowTypes; ++t) pl += (t < et && code_of(L, t) == 1) ? 1 : 0;
                F.drop_pos = __builtin_amdgcn_readlane(L.base + pl, __builtin_amdgcn_readfirstlane(el));
                F.fast = 2;
                if (lane == el) set_code(L, et, 0);
            }
            else if (ek == EV_COMPL) { if (lane == el) set_code(L, et, 3); }
            else if (ek == EV_DROPH) { if (lane == el) set_code(L, et, 0); finished = true; }
            else if (ek == EV_CAPIN) { if (lane == el) set_code(L, R_HWP, 3); finished = true; }
            else if (ek == EV_CAP) {
                int fin = 0;
                if (lane == el) {
                    const int g2 = et;
                    // members (code 1) of the group with their multipliers after the step
                    int bestm = -1; double bl = -1e300;
                    int pos = L.base;
#pragma unroll
                    for (int t = 0; t < kNumRowTypes; ++t) {
                        if (code_of(L, t) != 1) continue;
                        const double l = M.lam[pos] - tstep * M.rv[pos];
                        ++pos;
                        if (t >= R_AMAX && t <= R_VINC && group_of(t) == g2 && l > bl) { bl = l; bestm = t; }
                    }
                    const int p = pivot_of(L, g2);
                    const bool q_row_here = (kq == el) && !q.is_bound && q.gq == g2;
                    const bool q_bound_here = (kq == el) && q.is_bound && q.gq == g2;
                    if (p < 0) {                       // Z -> P
                        if (bestm < 0) { set_code(L, q.tq, 2); fin = 1; }
                        else set_code(L, bestm, 2);
                    } else {                           // pivot multiplier reached zero
                        set_code(L, p, 0);
                        if (bestm >= 0) set_code(L, bestm, 2);
                        else if (q_row_here) { set_code(L, q.tq, 2); fin = 1; }
                        else if (q_bound_here) { fin = 1; }
                    }
                }
                if (__any(fin)) finished = true;
            }
        }
        if (finished) { have_q = false; lam_q = 0.0; q.al = q.be = q.ga = q.de = q.d = 0.0; }
        PTOC(L, 5);
    }
    if (st.status == 0 && m > 0) refine_primal(L, c, M, Hs, m, 0.0, 0, 0.0, 0.0, 0.0, 0.0, grad_total, 4, 1e-14);
    st.m = m;
    return st;
}

// ----------------------------------------------------------------------------------------------
// per-step set-up: estimator, bounds, right-hand sides (A2, A3, A4 of SURVEY.md section 8a)

struct StepIn { double s, v, a_prev, t0, s_tv, v_tv, a_tv_prev; };

// ABO/RunOpt_ABMPC.m:287-324
__device__ void force_allocation(const DevCfg& C, double s_meas, double v_meas, double a_qp,
                                 double& Fm, double& Fb, double& a_real) {
    double sn, cs;
    slope_trig(C, s_meas, sn, cs);
    double F_r = -C.zeta_a * v_meas * v_meas - C.c_r * C.m * C.g * cs - C.m * C.g * sn;
    double F_t_req = C.m * C.lambda * a_qp - F_r;
    double F_f_r_max = C.mu / C.L * (C.m * C.g * (C.L_f * cs + C.h_g * sn) +
                                     C.h_g * (C.zeta_a * v_meas * v_meas + C.lambda * C.m * a_qp));
    double F_f_tot_max = C.mu * C.m * C.g * cs;
    if (F_t_req < 0.0) {
        double F_m_min = (v_meas < C.omega_m_r / C.phi) ? -C.phi * C.T_m_max / C.eta_TF : -C.P_m_max / C.eta_TF / v_meas;
        double fm = fmax(fmax(F_t_req, F_m_min), -F_f_r_max);
        Fm = fm;
        Fb = fmax(F_t_req, -F_f_tot_max) - fm;
    } else {
        double F_m_max = (v_meas < C.omega_m_r / C.phi) ? C.phi * C.T_m_max * C.eta_TF : C.P_m_max * C.eta_TF / v_meas;
        Fm = fmin(fmin(F_t_req, F_m_max), F_f_r_max);
        Fb = 0.0;
    }
    a_real = (Fm + Fb + F_r) / C.m / C.lambda;
}

struct StepOut { double out[EEPACC_OUT_N]; int status, iters; };

// One ABMPC step for the wave's instance (ABO/RunOpt_ABMPC.m:193-329).  `code` carries the
// working set between steps (already shifted by the caller).
template <int MMAX, int NS>
__device__ __forceinline__ void ab_step(const DevCfg& C, WaveMem<MMAX, NS>& M, double* Hs, const StepIn& in,
                        unsigned long long& code, StepOut& so, double& s_pred, double& v_pred,
                        const double* predp, bool pred_in_lds) {
    Lane L;
#ifdef EEPACC_AB_TIMING
    for (int i = 0; i < 14; ++i) L.prof[i] = 0;
    long long _t0 = wall_clock64();
#endif
    L.lane = lane_id(); L.N = C.N;
    const int lane = L.lane, N = C.N;
    const int kk = lane <= N ? lane : N;
    L.T = lane < N ? C.Tvec[lane] : 0.0;
    L.tau = C.tau[kk];
    L.tau1 = C.tau[kk + (lane < N ? 1 : 0)];
    L.tau_rev = C.tau[lane <= N ? N - lane : 0];
    Cfg c;
    c.N = N; c.tau_min = C.tau_min; c.wF = C.w_f; c.wS = C.w_s; c.wV = C.w_v; c.wH = 100.0 * C.w_h; c.qH = 2.0 * C.w_h;
    // estimators (A2)
    double s_est, v_est, stv_est, vtv_est;
    if (C.paramEstSetting == 2) {
        // EstimateVehicleTrajectory.m:81-88: shifted previous solution, [x_curr; prev(3:end); prev(end) + Ts v_prev(end)]
        // (previous z(1:7:end), z(2:7:end) kept in LDS between the steps of a launch, in C.pred across launches)
        const int idx = lane < N ? lane + 1 : N;
        const double ps = pred_in_lds ? M.ws[idx] : predp[idx], pv = pred_in_lds ? M.wv[idx] : predp[64 + idx];
        s_est = lane == 0 ? in.s : (lane < N ? ps : ps + C.Tvec[N - 1] * pv);
        v_est = lane == 0 ? in.v : pv;
        WSYNC();
    } else {
        estimate_traj(C, C.paramEstSetting, C.tConstACC_ego, in.s, in.v, in.a_prev, lane, s_est, v_est);
    }
    estimate_traj(C, C.TVestSetting, C.tConstACC_tar, in.s_tv, in.v_tv, in.a_tv_prev, lane, stv_est, vtv_est);
    const double dist_hor = bcast(s_est, N) - in.s;                          // :200
    const double stv_Nm1 = bcast(stv_est, N - 1);
    // bounds (A3)
    double v_lim, v_curv, v_stop, v_TL, a_min, a_max, j_min, j_max;
    route_bounds(C, s_est, v_est, in.t0, lane < N ? lane : N - 1, v_lim, v_curv, v_stop, v_TL, a_min, a_max, j_min, j_max);
    const double T_hwp = 2.0, A_hwp = 2.0, G_hwp = -0.0246 * T_hwp + 0.010819;
    L.chw = T_hwp + G_hwp * v_est;
    // free response of the double integrator and a-space right-hand sides (A4 + A5)
    const double sf = in.s + L.tau * in.v, vf = in.v;
    double b[kNumRowTypes];
    b[R_SLO] = -0.0; b[R_SHI] = C.s_goal; b[R_VLO] = -0.0; b[R_VHI] = C.v_max;
    b[R_AMAX] = a_max; b[R_AMIN] = -a_min;
    b[R_JMAX] = L.T * j_max + (lane == 0 ? in.a_prev : 0.0);
    b[R_JMIN] = -(L.T * j_min + (lane == 0 ? in.a_prev : 0.0));
    b[R_VLIM] = v_lim; b[R_VCURV] = v_curv; b[R_VSTOP] = v_stop; b[R_VTL] = v_TL;
    b[R_VINC] = -fmin(v_lim, v_curv);
    b[R_SAFE1] = stv_est - C.h_min; b[R_SAFE2] = stv_est; b[R_HWP] = stv_est - A_hwp;
    if (lane == N) { b[R_SAFE1] = stv_Nm1 - C.h_min; b[R_SAFE2] = stv_Nm1; }
    unsigned valid = 0u;
    L.lbF = L.lbS = L.lbV = L.lbH = 0.0;
    int infeasible_const = 0;
#pragma unroll
    for (int t = 0; t < kNumRowTypes; ++t) {
        double al = row_al(t), be = row_be(t, c.tau_min, L.chw);
        L.ba[t] = b[t] - al * sf - be * vf;
        bool exists;
        if (lane < N) {
            exists = true;
            if (t == R_SHI && !(C.s_goal < 1e300)) exists = false;
            if ((t == R_VLIM || t == R_VCURV || t == R_VSTOP || t == R_VTL) && !C.ab_route_rows) exists = false;
        } else exists = (lane == N) && (t == R_SAFE1 || t == R_SAFE2);
        if (exists && lane == 0 && row_ga(t) == 0.0) {
            // stage-0 rows without an a-component are constants: fold into slack bounds
            exists = false;
            int g2 = group_of(t);
            double need = -L.ba[t];
            if (g2 == G_NONE) { if (need > 1e-9) infeasible_const = 1; }
            else if (g2 == G_F) L.lbF = fmax(L.lbF, need);
            else if (g2 == G_S) L.lbS = fmax(L.lbS, need);
            else if (g2 == G_V) L.lbV = fmax(L.lbV, need);
            else L.lbH = fmax(L.lbH, need);
        }
        if (exists) valid |= (1u << t);
    }
    L.valid = valid;
    L.ign = 0u;
    // drop warm-start codes of rows that do not exist at this lane
    L.code = code;
#pragma unroll
    for (int t = 0; t < kNumRowTypes; ++t)
        if (!((valid >> t) & 1u)) set_code(L, t, 0);
    // base gradient (condensed objective, CreateQP_AB.m:162-180 through Psi'):
    //   g_j = T_j * (N-1-j) * (2 cq v0 + w_FC p10) + w_FC p01 lambda m ; g_0 -= 2 w_j/T_0 a_{-1}
    L.g0 = 0.0;
    if (lane < N) {
        L.g0 = L.T * (double)(N - 1 - lane) * (2.0 * C.cq * in.v + C.glin_v) + C.glin_a;
        if (lane == 0) L.g0 -= 2.0 * C.w_j / L.T * in.a_prev;
    }
    L.a = L.sh = L.vh = L.am1 = 0.0;
    double grad_total = 0.0;
    // per-wave inverse of the effective Hessian: start from the step-invariant H^-1
    he_load_base<NS>(Hs, C.Hinv, N, lane);
    L.kmask = 0ull;
    WSYNC();
#ifdef EEPACC_AB_TIMING
    L.prof[6] += wall_clock64() - _t0;
    _t0 = wall_clock64();
#endif
    SolveStats st = solve_qp<MMAX, NS>(L, c, M, Hs, C.Hinv, C.tau, C.max_iter, grad_total);
#ifdef EEPACC_AB_TIMING
    L.prof[8] += wall_clock64() - _t0;
    _t0 = wall_clock64();
#endif
    code = L.code;
    // recover z = Psi x + d (A7): predicted states
    s_pred = sf + L.sh; v_pred = vf + L.vh;
    // stage-0 slacks and the dense-QP objective value
    double xiF = 0, xiS = 0, xiV = 0, xiH = 0;
    if (lane < N) {
        xiF = fmax(group_xi(L, c, G_F), L.lbF);
        xiS = fmax(group_xi(L, c, G_S), L.lbS);
        xiV = fmax(group_xi(L, c, G_V), L.lbV);
        xiH = (code_of(L, R_HWP) == 3) ? fmax(row_val(L, c, R_HWP, L.ba[R_HWP]), L.lbH) : L.lbH;
    }
    // 1/2 a'Ha + g'a with H a = -(grad_total - g0) - g0 ... : H a = -grad_total  => a'(g0 - grad/2)
    // 1/2 a'Ha = 1/2 a'H_eff a - q/2 sum_K (n_k'a)^2 and H_eff a = -grad_total
    double part = (lane < N) ? L.a * (L.g0 - 0.5 * grad_total) : 0.0;
    part += C.w_f * xiF + C.w_s * xiS + C.w_v * xiV + c.wH * xiH + 0.5 * c.qH * xiH * xiH;
    if (lane < N && code_of(L, R_HWP) == 3) {
        const double na = row_val(L, c, R_HWP, L.ba[R_HWP]) + L.ba[R_HWP];      // n_k'a
        part -= 0.5 * c.qH * na * na;
    }
    const double cost = wave_sum(part);
    const double a0 = bcast(L.a, 0);
    double Fm, Fb, a_real;
    force_allocation(C, in.s, in.v, a0, Fm, Fb, a_real);
    so.out[EEPACC_OUT_S] = in.s;
    so.out[EEPACC_OUT_V] = in.v;
    so.out[EEPACC_OUT_FM] = Fm;
    so.out[EEPACC_OUT_FB] = Fb;
    so.out[EEPACC_OUT_A] = a_real;
    so.out[EEPACC_OUT_XI_V] = bcast(xiV, 0);
    so.out[EEPACC_OUT_XI_H] = bcast(xiH, 0);
    so.out[EEPACC_OUT_XI_S] = bcast(xiS, 0);
    so.out[EEPACC_OUT_XI_F] = bcast(xiF, 0);
    so.out[EEPACC_OUT_COST] = cost;
    so.out[EEPACC_OUT_DISTHOR] = dist_hor;
    so.out[EEPACC_OUT_AQP] = a0;
    so.status = (st.status != 0 || __any(infeasible_const)) ? 1 : 0;
#ifdef EEPACC_DEBUG_STATUS
    if (so.status) so.status = st.status * 100000 + (__any(infeasible_const) ? 10000 : 0) + (st.m + 100) + 1000 * 0;
#endif
#ifdef EEPACC_DEBUG_STATUS
    so.iters = st.iters + 100000 * st.status + 1000000 * (__any(infeasible_const) ? 1 : 0) + 10000000 * st.m;
#else
    so.iters = st.iters;
#endif
#ifdef EEPACC_AB_TIMING
    L.prof[7] += wall_clock64() - _t0;
    if (L.lane == 0)
        for (int i = 0; i < 14; ++i) atomicAdd(&g_ab_prof[i], (unsigned long long)L.prof[i]);
#endif
}

// receding-horizon shift of the working set: stage k takes stage k+1's codes, the last stage and
// the terminal rows keep theirs
__device__ __forceinline__ unsigned long long shift_codes(unsigned long long code, int N) {
    const int lane = lane_id();
    unsigned lo = (unsigned)code, hi = (unsigned)(code >> 32);
    unsigned nlo = __shfl_down(lo, 1, 64), nhi = __shfl_down(hi, 1, 64);
    unsigned long long nxt = ((unsigned long long)nhi << 32) | nlo;
    if (lane < N - 1) return nxt;
    return code;
}


// LDS layout of a block: per wave [WaveMem][He: N x N doubles]
__host__ __device__ inline size_t wave_bytes(size_t wm, int ns) {
    return ((wm + (size_t)ns * ns * sizeof(double)) + 15) & ~(size_t)15;
}

template <int MMAX, int NS>
__device__ WaveMem<MMAX, NS>* wave_mem(unsigned char* smem, int N, double*& He) {
    unsigned char* base = smem + wave_bytes(sizeof(WaveMem<MMAX, NS>), NS) * (threadIdx.x >> 6);
    He = reinterpret_cast<double*>(base + sizeof(WaveMem<MMAX, NS>));
    return reinterpret_cast<WaveMem<MMAX, NS>*>(base);
}

// B2: one step for B instances.  state: per instance 64 x uint64 codes (instance-major).
template <int MMAX, int NS, int WPB>
__global__ void __launch_bounds__(64 * WPB, ((NS <= 32 && WPB >= 4) ? 2 : 1))
k_ab_step(const DevCfg* __restrict__ Cp, int B,
          const double* __restrict__ s, const double* __restrict__ v, const double* __restrict__ a_prev,
          const double* __restrict__ t0, const double* __restrict__ s_tv, const double* __restrict__ v_tv,
          const double* __restrict__ a_tv_prev, unsigned long long* __restrict__ codes,
          double* __restrict__ out, double* __restrict__ s_pred, double* __restrict__ v_pred,
          int32_t* __restrict__ status, int32_t* __restrict__ iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevCfg& C = *Cp;
    rc_table_init<MMAX>();
    const int b = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (b >= B) return;
    double* Hs;
    WaveMem<MMAX, NS>& M = *wave_mem<MMAX, NS>(smem, C.N, Hs);
    const int lane = lane_id();
    StepIn in{s[b], v[b], a_prev[b], t0[b], s_tv[b], v_tv[b], a_tv_prev[b]};
    unsigned long long code = codes[(size_t)b * 64 + lane];
    StepOut so;
    double sp, vp;
    double* predp = C.pred + (size_t)b * 128;
    ab_step<MMAX, NS>(C, M, Hs, in, code, so, sp, vp, predp, false);
    if (C.paramEstSetting == 2 && lane <= C.N) { predp[lane] = sp; predp[64 + lane] = vp; }
    codes[(size_t)b * 64 + lane] = shift_codes(code, C.N);
    if (lane < EEPACC_OUT_N) {
        double val = 0.0;
#pragma unroll
        for (int f = 0; f < EEPACC_OUT_N; ++f) if (f == lane) val = so.out[f];
        out[(size_t)lane * B + b] = val;
    }
    if (s_pred && lane <= C.N) s_pred[(size_t)lane * B + b] = sp;
    if (v_pred && lane <= C.N) v_pred[(size_t)lane * B + b] = vp;
    if (lane == 0) { status[b] = so.status; if (iters) iters[b] = so.iters; }
}

// B1: closed loop over n_steps for B instances (ABO/RunOpt_ABMPC.m:154-340).  k_start > 0
// resumes from the carried per-instance state (carry [6][B]: s, v, Fm, Fb of the previous step,
// previous lead speed, t_0; codes: shifted working set).
template <int MMAX, int NS, int WPB>
__global__ void __launch_bounds__(64 * WPB, ((NS <= 32 && WPB >= 4) ? 2 : 1))
k_run_abmpc(const DevCfg* __restrict__ Cp, int B, int k_start, int n_steps,
            const double* __restrict__ s0, const double* __restrict__ v0, const double* __restrict__ a_m1,
            const double* __restrict__ s_tv, const double* __restrict__ v_tv,
            double* __restrict__ carry, unsigned long long* __restrict__ codes,
            double* __restrict__ traj, int32_t* __restrict__ status, int32_t* __restrict__ iters_total,
            int* __restrict__ work_counter, int* __restrict__ done, int kChunkSteps) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevCfg& C = *Cp;
    rc_table_init<MMAX>();
    double* Hs;
    WaveMem<MMAX, NS>& M = *wave_mem<MMAX, NS>(smem, C.N, Hs);
    const int lane = lane_id();
    const double Ts = C.Tvec[0];
    // Instances take very different numbers of working-set changes (per-instance run times spread
    // 0.75x..1.7x around the mean), so the simulation is cut into work units (instance, chunk of
    // kChunkSteps MPC steps) handed out through a device-wide counter in chunk-major order.  The loop
    // state of an instance travels between units through `carry`/`codes` in HBM: the producer wave
    // publishes done[b] = chunk+1 behind an agent-scope release, the consumer polls done[b] relaxed
    // and then takes one agent-scope acquire (cdna_hip_programming.md, Guideline 16).  A unit is only
    // handed out after its predecessor has been picked by a running wave, so the wait is bounded.
    const int n_chunks = (n_steps + kChunkSteps - 1) / kChunkSteps;
    const int n_units = n_chunks * B;
    for (int fetch = 0; fetch <= n_units; ++fetch) {
    int u = 0;
    if (lane == 0) u = atomicAdd(work_counter, 1);
    u = __builtin_amdgcn_readfirstlane(u);
    if (u >= n_units || u < 0) break;
    const int chunk = u / B, b = u - chunk * B;
    const int kk0 = chunk * kChunkSteps;
    const int kk1 = (kk0 + kChunkSteps < n_steps) ? kk0 + kChunkSteps : n_steps;
    if (chunk > 0) {
        int spins = 0;
        while (__hip_atomic_load(&done[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunk) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1 << 26)) break;       // never expected; keeps every wave finite
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef EEPACC_DEBUG_TIMING
    const long long t_begin = wall_clock64();
#endif
    unsigned long long code = 0ull;
    double s_prev = 0, v_prev = 0, Fm_prev = 0, Fb_prev = 0, v_tv_measured = 0.0, t_0 = 0.0;
    if (k_start + kk0 > 0) {
        s_prev = carry[0 * (size_t)B + b]; v_prev = carry[1 * (size_t)B + b];
        Fm_prev = carry[2 * (size_t)B + b]; Fb_prev = carry[3 * (size_t)B + b];
        v_tv_measured = carry[4 * (size_t)B + b]; t_0 = carry[5 * (size_t)B + b];
        code = codes[(size_t)b * 64 + lane];
    }
    int it_total = 0;
    double* predp = C.pred + (size_t)b * 128;
    for (int kk = kk0; kk < kk1; ++kk) {
        StepIn in;
        if (k_start + kk == 0) {                             // :159-172
            in.s = s0[b]; in.v = v0[b]; in.a_prev = a_m1[b];
            in.s_tv = s_tv[b]; in.v_tv = 0.0; in.a_tv_prev = 0.0;
            v_tv_measured = 0.0;
        } else {                                             // :173-191
            double sm, vm;
            plant_rk4(C, s_prev, v_prev, Fm_prev + Fb_prev, sm, vm);
            in.s = sm; in.v = vm;
            in.a_prev = (vm - v_prev) / Ts;
            in.s_tv = s_tv[(size_t)kk * B + b];
            double v_tv_prev = v_tv_measured;
            v_tv_measured = v_tv[(size_t)kk * B + b];
            in.v_tv = v_tv_measured;
            in.a_tv_prev = (v_tv_measured - v_tv_prev) / Ts;
        }
        in.t0 = t_0;
        StepOut so;
        double sp, vp;
        ab_step<MMAX, NS>(C, M, Hs, in, code, so, sp, vp, predp, kk > kk0);
        if (C.paramEstSetting == 2) {
            WSYNC();
            if (lane <= C.N) { M.ws[lane] = sp; M.wv[lane] = vp; }
            WSYNC();
            if (kk == kk1 - 1 && lane <= C.N) { predp[lane] = sp; predp[64 + lane] = vp; }
        }
        code = shift_codes(code, C.N);
        if (lane < EEPACC_OUT_N) {
            double val = 0.0;
#pragma unroll
            for (int f = 0; f < EEPACC_OUT_N; ++f) if (f == lane) val = so.out[f];
            traj[((size_t)kk * EEPACC_OUT_N + lane) * B + b] = val;
        }
        if (lane == 0) status[(size_t)kk * B + b] = so.status;
        it_total += so.iters;
        s_prev = so.out[EEPACC_OUT_S]; v_prev = so.out[EEPACC_OUT_V];
        Fm_prev = so.out[EEPACC_OUT_FM]; Fb_prev = so.out[EEPACC_OUT_FB];
        t_0 += Ts;                                           // :329
    }
    codes[(size_t)b * 64 + lane] = code;
    if (lane == 0) {
        carry[0 * (size_t)B + b] = s_prev; carry[1 * (size_t)B + b] = v_prev;
        carry[2 * (size_t)B + b] = Fm_prev; carry[3 * (size_t)B + b] = Fb_prev;
        carry[4 * (size_t)B + b] = v_tv_measured; carry[5 * (size_t)B + b] = t_0;
#ifdef EEPACC_DEBUG_TIMING
        if (iters_total) atomicAdd(&iters_total[b], (int)((wall_clock64() - t_begin) / 100));   // microseconds
#else
        if (iters_total) atomicAdd(&iters_total[b], it_total);
#endif
    }
    // publish the unit: all of this wave's stores, then release, then the flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&done[b], chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }
}

// A10: post-processing (ABO/RunOpt_ABMPC.m:343-349), one thread per instance, sequential in time
__global__ void k_postprocess(const DevCfg* __restrict__ Cp, int B, int n_steps, const double* __restrict__ traj,
                              double* __restrict__ rpm, double* __restrict__ Tm, double* __restrict__ P,
                              double* __restrict__ E) {
    const DevCfg& C = *Cp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double Ts = C.Tvec[0];
    double acc = 0.0;
    const double kr = (30.0 / 3.14159265358979323846);
    for (int k = 0; k < n_steps; ++k) {
        const double v = traj[((size_t)k * EEPACC_OUT_N + EEPACC_OUT_V) * B + b];
        const double x = traj[((size_t)k * EEPACC_OUT_N + EEPACC_OUT_FM) * B + b];
        const double y = kr * v * C.phi;
        const double sg = (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0);
        const double tm = x / C.phi / pow(C.eta_TF, sg);
        const double* bb = C.b5;
        const double x2 = x * x, x3 = x2 * x, x4 = x3 * x, x5 = x4 * x;
        const double y2 = y * y, y3 = y2 * y, y4 = y3 * y, y5 = y4 * y;
        const double p = bb[0] + bb[1] * x + bb[2] * y + bb[3] * x2 + bb[4] * x * y + bb[5] * y2 + bb[6] * x3 +
                         bb[7] * x2 * y + bb[8] * x * y2 + bb[9] * y3 + bb[10] * x4 + bb[11] * x3 * y +
                         bb[12] * x2 * y2 + bb[13] * x * y3 + bb[14] * y4 + bb[15] * x5 + bb[16] * x4 * y +
                         bb[17] * x3 * y2 + bb[18] * x2 * y3 + bb[19] * x * y4 + bb[20] * y5;
        acc += p;
        const size_t o = (size_t)k * B + b;
        rpm[o] = y; Tm[o] = tm; P[o] = p; E[o] = Ts * acc;
    }
}

}  // namespace eepacc

// ----------------------------------------------------------------------------------------------
// host-side launchers used by eepacc_capi.cpp
namespace eepacc {

// working-set capacity: rigid rows are linearly independent, so m <= N (+ terminal rows)
constexpr int kMMaxSmall = 34, kNSSmall = 32;     // N <= 32: 8 waves / CU
constexpr int kMMaxLarge = 66, kNSLarge = 64;     // N <= 63: 2 waves / CU

static int waves_per_block() {
    static int w = -1;
    if (w < 0) {
        const char* e = getenv("EEPACC_WPB");
        w = (e && atoi(e) == 1) ? 1 : 4;
    }
    return w;
}

#ifdef EEPACC_AB_TIMING
extern "C" int eepacc_debug_ab_prof(unsigned long long* out, int reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ab_prof), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_ab_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

size_t ab_smem_bytes(int N) {
    return N <= kNSSmall ? wave_bytes(sizeof(WaveMem<kMMaxSmall, kNSSmall>), kNSSmall) * waves_per_block()
                         : wave_bytes(sizeof(WaveMem<kMMaxLarge, kNSLarge>), kNSLarge) * 2;
}

#define EEPACC_LAUNCH(KERNEL, MM, NSV, WPB, ...)                                                            \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<MM, NSV, WPB>), dim3((B + WPB - 1) / WPB), dim3(64 * WPB),  \
                       ab_smem_bytes(N), stream, __VA_ARGS__)

hipError_t launch_ab_step(const DevCfg* dC, int N, int B, const double* s, const double* v, const double* a_prev,
                          const double* t0, const double* s_tv, const double* v_tv, const double* a_tv_prev,
                          unsigned long long* codes, double* out, double* s_pred, double* v_pred,
                          int32_t* status, int32_t* iters, hipStream_t stream) {
    if (N > kNSSmall) EEPACC_LAUNCH(k_ab_step, kMMaxLarge, kNSLarge, 2, dC, B, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, codes, out, s_pred, v_pred, status, iters);
    else if (waves_per_block() == 1) EEPACC_LAUNCH(k_ab_step, kMMaxSmall, kNSSmall, 1, dC, B, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, codes, out, s_pred, v_pred, status, iters);
    else EEPACC_LAUNCH(k_ab_step, kMMaxSmall, kNSSmall, 4, dC, B, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, codes, out, s_pred, v_pred, status, iters);
    return hipGetLastError();
}

#define EEPACC_LAUNCH_GRID(KERNEL, MM, NSV, WPB, GRID, ...)                                               \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<MM, NSV, WPB>), dim3(GRID), dim3(64 * WPB), ab_smem_bytes(N), stream, __VA_ARGS__)

hipError_t launch_run_abmpc(const DevCfg* dC, int N, int B, int k_start, int n_steps, const double* s0,
                            const double* v0, const double* a_m1, const double* s_tv, const double* v_tv,
                            double* carry, unsigned long long* codes, double* traj,
                            int32_t* status, int32_t* iters_total, int* work_counter, int* done, int num_cus,
                            hipStream_t stream) {
    hipError_t e = hipMemsetAsync(work_counter, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(done, 0, sizeof(int) * (size_t)B, stream);
    if (e != hipSuccess) return e;
    if (iters_total) {
        e = hipMemsetAsync(iters_total, 0, sizeof(int32_t) * (size_t)B, stream);
        if (e != hipSuccess) return e;
    }
    static int kChunkSteps = -1;
    if (kChunkSteps < 0) {
        const char* ev = getenv("EEPACC_CHUNK");
        kChunkSteps = (ev && atoi(ev) > 0) ? atoi(ev) : kChunkStepsDefault;
    }
    const int n_units = ((n_steps + kChunkSteps - 1) / kChunkSteps) * B;
    // one chip-filling wave of blocks: LDS admits 8 (small) / 2 (large) waves per CU
    if (N > kNSSmall) {
        int grid = num_cus * 1, need = (n_units + 1) / 2;
        if (grid > need) grid = need;
        EEPACC_LAUNCH_GRID(k_run_abmpc, kMMaxLarge, kNSLarge, 2, grid, dC, B, k_start, n_steps, s0, v0, a_m1, s_tv, v_tv, carry, codes, traj, status, iters_total, work_counter, done, kChunkSteps);
    } else if (waves_per_block() == 1) {
        int grid = num_cus * 8; if (grid > n_units) grid = n_units;
        EEPACC_LAUNCH_GRID(k_run_abmpc, kMMaxSmall, kNSSmall, 1, grid, dC, B, k_start, n_steps, s0, v0, a_m1, s_tv, v_tv, carry, codes, traj, status, iters_total, work_counter, done, kChunkSteps);
    } else {
        int grid = num_cus * 2, need = (n_units + 3) / 4;
        if (grid > need) grid = need;
        EEPACC_LAUNCH_GRID(k_run_abmpc, kMMaxSmall, kNSSmall, 4, grid, dC, B, k_start, n_steps, s0, v0, a_m1, s_tv, v_tv, carry, codes, traj, status, iters_total, work_counter, done, kChunkSteps);
    }
    return hipGetLastError();
}

hipError_t launch_postprocess(const DevCfg* dC, int B, int n_steps, const double* traj, double* rpm, double* Tm,
                              double* P, double* E, hipStream_t stream) {
    hipLaunchKernelGGL(k_postprocess, dim3((B + 127) / 128), dim3(128), 0, stream, dC, B, n_steps, traj, rpm, Tm, P, E);
    return hipGetLastError();
}

hipError_t set_max_smem() {
    const void* fns[6] = {reinterpret_cast<const void*>(&k_ab_step<kMMaxSmall, kNSSmall, 1>),
                          reinterpret_cast<const void*>(&k_ab_step<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&k_ab_step<kMMaxLarge, kNSLarge, 2>),
                          reinterpret_cast<const void*>(&k_run_abmpc<kMMaxSmall, kNSSmall, 1>),
                          reinterpret_cast<const void*>(&k_run_abmpc<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&k_run_abmpc<kMMaxLarge, kNSLarge, 2>)};
    for (int i = 0; i < 6; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);   // the rest holds the static index table
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace eepacc
