// eepacc_fbs.hip -- structured FBMPC kernels (gfx950): the per-step pipeline of ABO/RunOpt_FBMPC.m:161-331
// (measurement, estimators, bounds, CreateQP_FB, condensing with the carried A(k)/D(k), dense QP, extraction)
// fused into one wavefront per instance, without ever forming the reference's dense QP.
//
// Formulation (numpy model: tools/proto_fb_structured.py).  Per stage k the reference has
// Fm_k, Fb_k and four slacks (ABO/Functions/MPCs/CreateQP_FB.m:158-489).  Here
//     u_k = Fm_k + Fb_k : total force.  The dynamics v_{k+1} = A22_k v_k + T_k/(lambda m) u_k + D2_k, the
//                         acceleration and jerk penalties and most rows see only u; dense N x N coupling,
//                         positive definite.
//     w_k = -Fb_k >= 0  : friction-brake share.  No curvature of its own; its price (c5 v_k + c2) is bilinear with
//                         the predicted speed (power term :181-184) -- that term is what makes the reference's
//                         dense Hessian indefinite.
// Every row is  al*s_k + be*v_k + ga*u_k + de*u_{k-1} + aw*w_k - xi_group <= b.  The slacks are eliminated exactly
// as in the ABMPC kernel (capped-multiplier groups, penalty for the quadratic slack).  w_k is treated like a slack
// with a state-dependent price: on its bound (w = 0) the rows containing it are ordinary rows; off its bound it is
// defined by a pivot row (torque limit, rear-axle limit, motor-force bound) and its bilinear term is folded into
// the per-wave inverse Hessian by a rank-2 update.  Pinning w this way is the inertia control: the reduced Hessian
// stays positive definite by construction of the working set, no proximal regularisation.
//
// Condensing: with Pi_k = prod_{i<k} A22_i, gamma_i = beta_i / Pi_{i+1}, Theta_k = sum_{i<k} T_i Pi_i the
// sensitivities are  dv_k/du_i = Pi_k gamma_i,  ds_k/du_i = gamma_i (Theta_k - Theta_{i+1})  (i < k), so the
// forward response and the adjoint are plain wave prefix / suffix sums with per-lane scalings (DPP), and the
// entries of H = Psi' Q Psi have closed forms (built per step, inverted in LDS by symmetric sweeps).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "eepacc_device.h"
#include "eepacc_stage.h"
#include "eepacc_wave.h"
#include "eepacc_fbs.h"
#include "../../include/eepacc.h"

namespace eepacc {
namespace fbs {
using namespace wv;

#define WSYNC() EEPACC_WSYNC()

#ifdef EEPACC_FBS_TIMING
__device__ unsigned long long g_fbs_prof[16];
#define FT_DECL long long _ft = wall_clock64(); long long _fp[16] = {0}
#define FT_TOC(slot) do { long long _n = wall_clock64(); _fp[slot] += _n - _ft; _ft = _n; } while (0)
#define FT_FLUSH() do { if (lane_id() == 0) for (int _i = 0; _i < 16; ++_i) if (_fp[_i]) atomicAdd(&g_fbs_prof[_i], (unsigned long long)_fp[_i]); } while (0)
#else
#define FT_DECL
#define FT_TOC(slot)
#define FT_FLUSH()
#endif

constexpr double kInf = 1e300;
constexpr double kTolViol = 1e-11;
constexpr double kTolDual = 1e-12;
constexpr int kSinglePasses = 8;

// ----------------------------------------------------------------------------------------------
// row catalogue (order of the groups matters: types of one group are contiguous)
enum FRow : int {
    F_SLO = 0, F_SHI, F_VLO, F_VHI,                     // hard state bounds          CreateQP_FB.m:311-318
    F_FMLO, F_FMHI, F_FBLO,                             // hard force bounds (contain w)          :319-326
    F_TQMIN, F_TQMAX, F_RTLO, F_RTHI,                   // group F, contain w (torque, rear axle) :359-377
    F_FTHI, F_FTLO, F_AMAX, F_AMIN, F_JMAX, F_JMIN, F_VLIM, F_VCURV,   // group F           :380-426
    F_SAFE1, F_SAFE2, F_VSTOP, F_VTL,                   // group S                                :429-458
    F_VINC,                                             // group V                                :445-448
    F_HWP,                                              // group H, quadratic slack               :461-473
    kNumF = 25
};
enum FGroup : int { GN = 0, GF = 1, GS = 2, GV = 3, GH = 4, GW = 5 };
enum Ev : int { EV_NONE = 0, EV_DROP, EV_COMPL, EV_DROPH, EV_CAP, EV_CAPIN };

struct RC {        // wave-uniform row constants
    double tau_min, c1, g_tqmin, g_tqmax, g_rtlo, g_rthi, Lmu;
    double wF, wS, wV, wH, qH, c5, c2;
};

__device__ __forceinline__ int group_of(int t) {
    return t <= F_FBLO ? GN : (t <= F_VCURV ? GF : (t <= F_VTL ? GS : (t == F_VINC ? GV : GH)));
}
__device__ __forceinline__ double row_al(int t) {
    return t == F_SLO ? -1.0 : ((t == F_SHI || t == F_SAFE1 || t == F_SAFE2 || t == F_HWP) ? 1.0 : 0.0);
}
__device__ __forceinline__ double row_be(int t, const RC& c, double chw) {
    switch (t) {
        case F_VLO: case F_VINC: return -1.0;
        case F_VHI: case F_VLIM: case F_VCURV: case F_VSTOP: case F_VTL: return 1.0;
        case F_TQMIN: case F_TQMAX: return c.c1;
        case F_SAFE2: return c.tau_min;
        case F_HWP: return chw;
        default: return 0.0;
    }
}
__device__ __forceinline__ double row_ga(int t, const RC& c) {
    switch (t) {
        case F_FMHI: case F_FTHI: case F_AMAX: case F_JMAX: return 1.0;
        case F_FMLO: case F_FTLO: case F_AMIN: case F_JMIN: return -1.0;
        case F_TQMIN: return c.g_tqmin;
        case F_TQMAX: return c.g_tqmax;
        case F_RTLO: return c.g_rtlo;
        case F_RTHI: return c.g_rthi;
        default: return 0.0;
    }
}
__device__ __forceinline__ double row_de(int t, int k) {
    if (k == 0) return 0.0;
    return t == F_JMAX ? -1.0 : (t == F_JMIN ? 1.0 : 0.0);
}
// coefficient of w_k = -Fb_k (Fm = u + w):  < 0 : w relaxes the row, > 0 : w tightens it
__device__ __forceinline__ double row_aw(int t, const RC& c) {
    switch (t) {
        case F_FMLO: return -1.0;
        case F_FMHI: case F_FBLO: return 1.0;
        case F_TQMIN: return c.g_tqmin;
        case F_TQMAX: return c.g_tqmax;
        case F_RTLO: return -c.Lmu;
        case F_RTHI: return c.Lmu;
        default: return 0.0;
    }
}
// Right-hand sides that are equal by construction share a slot of FMem::ba: the three force bounds (1e4 N), the two
// torque limits, the two rear-axle limits, the two total-friction limits (CreateQP_FB.m:319-326, 359-387).
constexpr unsigned kSlotOwner = 0x1ffffffu & ~((1u << F_FMHI) | (1u << F_FBLO) | (1u << F_TQMAX) | (1u << F_RTHI) | (1u << F_FTLO));
constexpr int kNumSlots = 20;
static_assert(__builtin_popcount(kSlotOwner) == kNumSlots, "slot map of the right-hand sides");
__device__ __forceinline__ constexpr int slot_of(int t) { return __builtin_popcount(kSlotOwner & ((2u << t) - 1u)) - 1; }
__device__ __forceinline__ constexpr bool owns_slot(int t) { return ((kSlotOwner >> t) & 1u) != 0u; }
__device__ __forceinline__ bool is_wrow(int t) { return t >= F_FMLO && t <= F_RTHI; }
__device__ __forceinline__ bool is_relax(int t) { return t == F_FMLO || t == F_TQMIN || t == F_RTLO; }

// LDS diet of round 3 (7 instead of 6 waves per CU at N <= 32): the input vectors of the He products live in ws / wv
// (free outside gradient_side), the pivot column of the factor updates in wa, Pi_k / Theta_k are read from the lane that
// holds them, and row types with identical right-hand sides share a slot of `ba`.
template <int MMAX, int NS>
struct FMem {                 // one per wave, in LDS (followed by the wave's NS x NS matrix He)
    double P[MMAX * (MMAX + 1) / 2];
    double av[NS];
    double shv[NS + 1], vhv[NS + 1];
    double ub[NS + 1], sub[NS + 1], vub[NS + 1];      // images of a vector; also scratch of adjoint()
    double ws[NS + 1], wv[NS + 1], wa[NS + 1];        // stage weights of gradient_side; yv = ws, yv2 = wv, colk = wa otherwise
    double e_al[MMAX], e_be[MMAX], e_ga[MMAX], e_de[MMAX], e_d[MMAX];
    double lam[MMAX], sv[MMAX], rv[MMAX];
    double ba[kNumSlots * (NS + 1)];                  // right-hand sides, [slot of the type][lane]
    int w_k[MMAX];
};

__device__ __forceinline__ int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

struct Tup { double al, be, ga, de, c; };
__device__ __forceinline__ Tup tup0() { return Tup{0.0, 0.0, 0.0, 0.0, 0.0}; }

// per-lane (= per-stage) registers of the wave's QP
struct Lane {
    int lane, N;
    double T, gam, Pi, Th, Th1;   // T_k, gamma_k, Pi_k, Theta_k, Theta_{k+1}
    double vbar;                  // free response of the speed at this stage
    double chw;                   // headway-policy coefficient
    double g0;                    // base gradient of the condensed objective
    unsigned valid, ign;          // bit t: row exists / row ignored in this solve (bit 25+g: bound of group g)
    unsigned long long kmask;     // wave-uniform: stages whose xi_h penalty q n n' is folded into He
    unsigned long long wmask;     // wave-uniform: stages whose bilinear c5 v_k w_k term is folded into He
    double fal, fbe, fga, fde;    // gradient of the folded w_k expression (this lane)
    double lbF, lbS, lbV, lbH;    // slack lower bounds (constant rows of stage 0 fold in here)
    unsigned long long code;      // 2 bits per type: 0 off, 1 in working set, 2 pivot of its slack group,
                                  // 3 HWP: penalised / FMLO,TQMIN,RTLO: pivot of w
    double u, sh, vh, um1;        // total force, homogeneous trajectories, u_{k-1}
    int base;                     // first working-set position of this lane's rows
    int unsup;                    // state outside what this solver represents (reported as status 1)
};

__device__ __forceinline__ int code_of(const Lane& L, int t) { return (int)((L.code >> (2 * t)) & 3ull); }
__device__ __forceinline__ void set_code(Lane& L, int t, int c) {
    L.code = (L.code & ~(3ull << (2 * t))) | ((unsigned long long)c << (2 * t));
}
__device__ __forceinline__ int lane_group(const Lane& L, int t) { return L.lane == L.N ? GN : group_of(t); }
__device__ __forceinline__ double group_w(const RC& c, int g) { return g == GF ? c.wF : (g == GS ? c.wS : (g == GV ? c.wV : c.wH)); }
__device__ __forceinline__ double group_lb(const Lane& L, int g) { return g == GF ? L.lbF : (g == GS ? L.lbS : (g == GV ? L.lbV : L.lbH)); }

template <int NS> __device__ __forceinline__ double ba_of(const double* ba, int t, int lane) { return ba[slot_of(t) * (NS + 1) + lane]; }

// bit tricks on the 2-bit codes: bit 2t of the result is set where type t has the given code
constexpr unsigned long long kEven = 0x5555555555555555ull;
__device__ __forceinline__ unsigned long long codes_eq1(unsigned long long c) { return c & ~(c >> 1) & kEven; }
__device__ __forceinline__ unsigned long long codes_eq2(unsigned long long c) { return (c >> 1) & ~c & kEven; }
__device__ __forceinline__ unsigned long long codes_eq3(unsigned long long c) { return (c >> 1) & c & kEven; }
__device__ __forceinline__ constexpr unsigned long long type_range(int t0, int t1) {      // even bits of types t0..t1
    return (((t1 >= 31) ? ~0ull : ((1ull << (2 * (t1 + 1))) - 1ull)) & ~((1ull << (2 * t0)) - 1ull)) & kEven;
}
// pivot type of linear group g at this lane (-1: the group's slack is on its bound)
__device__ __forceinline__ int pivot_of(const Lane& L, int g) {
    const unsigned long long rng = g == GF ? type_range(F_TQMIN, F_VCURV) : (g == GS ? type_range(F_SAFE1, F_VTL) : type_range(F_VINC, F_VINC));
    const unsigned long long two = codes_eq2(L.code) & rng;
    return two ? ((__ffsll((long long)two) - 1) >> 1) : -1;
}
__device__ __forceinline__ int wpivot_of(const Lane& L) {
    if (L.lane >= L.N) return -1;
    const unsigned long long three = codes_eq3(L.code) & ((1ull << (2 * F_FMLO)) | (1ull << (2 * F_FBLO)) | (1ull << (2 * F_TQMIN)) | (1ull << (2 * F_RTLO)));
    return three ? ((__ffsll((long long)three) - 1) >> 1) : -1;
}
// the pivot of w is one of the rows that also carry xi_f (torque / rear-axle limit): then xi_f must be defined first,
// by a row free of w.  Otherwise (motor-force bound, or w on its upper bound = pivot F_FBLO) w is defined first and
// the pivot of xi_f may contain it.
__device__ __forceinline__ bool wpivot_in_F(int p) { return p == F_TQMIN || p == F_RTLO; }

// homogeneous response to the per-lane input x (lane k < N holds x_k): vh_k = Pi_k sum_{i<k} gamma_i x_i,
// sh_k = sum_{i<k} T_i vh_i
__device__ __forceinline__ void hom_traj(const Lane& L, double x, double& sh, double& vh) {
    const double xi = (L.lane < L.N) ? L.gam * x : 0.0;
    vh = L.Pi * scan_excl(xi);
    sh = scan_excl((L.lane < L.N) ? L.T * vh : 0.0);
}

// out_k = sum_i He[i][k] * yv[i]   (He symmetric NS x NS in LDS, zero padded)
template <int NS>
__device__ __forceinline__ double hinv_mul(const double* Hs, const double* yv, int N, int lane) {
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

// u-space normal of the row (kq; al,be,ga,de) evaluated at this lane j (Pi_kq, Th_kq: wave-uniform)
__device__ __forceinline__ double normal_at(const Lane& L, int kq, double al, double be, double ga, double de,
                                            double Pi_kq, double Th_kq) {
    double c = 0.0;
    const int j = L.lane;
    if (j < L.N) {
        if (j < kq) c = L.gam * (be * Pi_kq + al * (Th_kq - L.Th1));
        if (j == kq) c += ga;
        if (j == kq - 1) c += de;
    }
    return c;
}

// adjoint of the condensing: stage weights on (s_k, v_k, u_k) in LDS (ws, wv, wa; k = 0..N) -> d/du_j for lane j
template <int MMAX, int NS>
__device__ __forceinline__ double adjoint(const Lane& L, const FMem<MMAX, NS>& M, const double* ws, const double* wv,
                                          const double* wa, double* tmp) {
    // suffix sums over stages k > j = prefix sums over the reversed stage order (lane r holds stage N - r)
    const int r = L.lane, N = L.N;
    const int k = N - r;
    const bool in = k >= 0;
    const double s = in ? ws[k] : 0.0;
    const double Pik = __shfl(L.Pi, in ? k : 0, 64), Thk = __shfl(L.Th, in ? k : 0, 64);      // Pi_k, Theta_k live on lane k
    const double x = in ? fma(Pik, wv[k], Thk * s) : 0.0;
    const double S0 = scan_excl(s), S1 = scan_excl(x);
    if (in) { tmp[k] = S0; tmp[(NS + 1) + k] = S1; }
    WSYNC();
    const int j = L.lane;
    double g = 0.0;
    if (j < N) g = wa[j] + L.gam * (tmp[(NS + 1) + j] - L.Th1 * tmp[j]);
    WSYNC();
    return g;
}

// ----------------------------------------------------------------------------------------------
// local variables of a stage as affine functions of (s_k, v_k, u_k, u_{k-1}) (homogeneous parts)
template <int NS>
__device__ __forceinline__ Tup xi_expr(const Lane& L, const RC& c, const double* ba, int g) {
    Tup X = tup0();
    if (L.lane >= L.N) return X;
    if (g == GH) {
        if (code_of(L, F_HWP) == 3) { X.al = 1.0; X.be = L.chw; X.c = -ba_of<NS>(ba, F_HWP, L.lane); }
        else X.c = L.lbH;
        return X;
    }
    const int p = pivot_of(L, g);
    if (p < 0) { X.c = group_lb(L, g); return X; }
    X.al = row_al(p); X.be = row_be(p, c, L.chw); X.ga = row_ga(p, c); X.de = row_de(p, L.lane);
    X.c = -ba_of<NS>(ba, p, L.lane);
    return X;
}
// w_k: zero on its bound, else defined by its pivot row q (aw_q < 0):  w = (row_q - xi_{g(q)} - b_q) / |aw_q|
template <int NS>
__device__ __forceinline__ Tup w_expr(const Lane& L, const RC& c, const double* ba, const Tup& XF) {
    Tup W = tup0();
    const int q = wpivot_of(L);
    if (q < 0) return W;
    const double ia = -1.0 / row_aw(q, c);
    const bool inF = wpivot_in_F(q);
    W.al = (row_al(q) - (inF ? XF.al : 0.0)) * ia;
    W.be = (row_be(q, c, L.chw) - (inF ? XF.be : 0.0)) * ia;
    W.ga = (row_ga(q, c) - (inF ? XF.ga : 0.0)) * ia;
    W.de = (row_de(q, L.lane) - (inF ? XF.de : 0.0)) * ia;
    W.c = (-ba_of<NS>(ba, q, L.lane) - (inF ? XF.c : 0.0)) * ia;
    return W;
}
struct Locals { Tup XF, XS, XV, XH, W; };
template <int NS>
__device__ __forceinline__ Locals locals_of(const Lane& L, const RC& c, const double* ba) {
    Locals S;
    S.XF = xi_expr<NS>(L, c, ba, GF); S.XS = xi_expr<NS>(L, c, ba, GS); S.XV = xi_expr<NS>(L, c, ba, GV);
    S.XH = xi_expr<NS>(L, c, ba, GH);
    S.W = w_expr<NS>(L, c, ba, S.XF);
    const int pW = wpivot_of(L);
    if (pW >= 0 && !wpivot_in_F(pW)) {
        // w first: a pivot of xi_f that contains w sees it through w's own expression
        const int pF = pivot_of(L, GF);
        const double aw = pF >= 0 ? row_aw(pF, c) : 0.0;
        if (aw != 0.0) { S.XF.al += aw * S.W.al; S.XF.be += aw * S.W.be; S.XF.ga += aw * S.W.ga; S.XF.de += aw * S.W.de; S.XF.c += aw * S.W.c; }
    }
    return S;
}
__device__ __forceinline__ const Tup& xi_of(const Locals& S, int g) { return g == GF ? S.XF : (g == GS ? S.XS : (g == GV ? S.XV : S.XH)); }
__device__ __forceinline__ double tup_val(const Tup& X, const Lane& L) {
    return X.al * L.sh + X.be * L.vh + X.ga * L.u + X.de * L.um1 + X.c;
}
// effective row of type t at this lane: row + aw*w(x) - xi_g(x) <= d
template <int NS>
__device__ __forceinline__ Tup eff_row(const Lane& L, const RC& c, const double* ba, const Locals& S, int t) {
    Tup R;
    R.al = row_al(t); R.be = row_be(t, c, L.chw); R.ga = row_ga(t, c); R.de = row_de(t, L.lane);
    R.c = ba_of<NS>(ba, t, L.lane);
    const double aw = row_aw(t, c);
    if (aw != 0.0) { R.al += aw * S.W.al; R.be += aw * S.W.be; R.ga += aw * S.W.ga; R.de += aw * S.W.de; R.c -= aw * S.W.c; }
    const int g = lane_group(L, t);
    if (g != GN) { const Tup& X = xi_of(S, g); R.al -= X.al; R.be -= X.be; R.ga -= X.ga; R.de -= X.de; R.c += X.c; }
    return R;
}

struct SolveStats { int status, iters, events, m; };

// ----------------------------------------------------------------------------------------------
// per-wave inverse He of the effective Hessian.  Rank-one: penalty q n n' of a stage whose xi_h is off its bound.
template <int MMAX, int NS>
__device__ __forceinline__ void he_rank1(const Lane& L, const RC& c, FMem<MMAX, NS>& M, double* He, int k, bool add) {
    const int lane = L.lane, N = L.N;
    const double chwk = bcast(L.chw, k);
    const double nk = normal_at(L, k, 1.0, chwk, 0.0, 0.0, bcast(L.Pi, k), bcast(L.Th, k));
    if (lane < NS) M.ws[lane] = nk;
    WSYNC();
    const double y = hinv_mul<NS>(He, M.ws, N, lane);
    double sy, vy;
    hom_traj(L, y, sy, vy);
    const double ny = bcast(sy + chwk * vy, k);
    const double kappa = add ? c.qH / (1.0 + c.qH * ny) : -c.qH / (1.0 - c.qH * ny);
    if (lane < NS) M.ub[lane] = y;
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

// Rank-two: H +- c5 (sigma_k f' + f sigma_k'), sigma_k = dv_k/du, f = gradient of the expression that defines w_k.
// Woodbury with U = [sigma, f], K = +-c5 [[0,1],[1,0]]:  He -= (He U) (K^-1 + U' He U)^-1 (He U)'.
template <int MMAX, int NS>
__device__ __forceinline__ bool he_rank2(const Lane& L, const RC& c, FMem<MMAX, NS>& M, double* He, int k,
                                         double fal, double fbe, double fga, double fde, bool add) {
    const int lane = L.lane, N = L.N;
    const double Pk = bcast(L.Pi, k), Tk = bcast(L.Th, k);
    const double sg = normal_at(L, k, 0.0, 1.0, 0.0, 0.0, Pk, Tk);
    const double fv = normal_at(L, k, fal, fbe, fga, fde, Pk, Tk);
    if (lane < NS) { M.ws[lane] = sg; M.wv[lane] = fv; }
    WSYNC();
    double y1, y2;
    hinv_mul2<NS>(He, M.ws, M.wv, N, lane, y1, y2);
    double s1, v1, s2, v2;
    hom_traj(L, y1, s1, v1);
    hom_traj(L, y2, s2, v2);
    const int km1 = k > 0 ? k - 1 : 0;
    const double g11 = bcast(v1, k), g12 = bcast(v2, k);
    const double g22 = fal * bcast(s2, k) + fbe * g12 + fga * bcast(y2, k) + (k > 0 ? fde * bcast(y2, km1) : 0.0);
    const double ik = (add ? 1.0 : -1.0) / c.c5;
    const double a11 = g11, a12 = ik + g12, a22 = g22;
    const double det = a11 * a22 - a12 * a12;
    if (!(fabs(det) > 1e-300)) return false;
    const double m11 = a22 / det, m12 = -a12 / det, m22 = a11 / det;
    if (lane < NS) { M.ub[lane] = y1; M.sub[lane] = y2; }
    WSYNC();
    if (lane < NS) {
        const double c1j = m11 * y1 + m12 * y2, c2j = m12 * y1 + m22 * y2;
        double* col = He + lane;
#pragma unroll
        for (int i = 0; i < NS; i += 2) {
            const double h0 = col[(i + 0) * NS], h1 = col[(i + 1) * NS];
            col[(i + 0) * NS] = fma(-M.sub[i], c2j, fma(-M.ub[i], c1j, h0));
            col[(i + 1) * NS] = fma(-M.sub[i + 1], c2j, fma(-M.ub[i + 1], c1j, h1));
        }
    }
    WSYNC();
    return true;
}

template <int MMAX, int NS>
__device__ __forceinline__ void he_sync(Lane& L, const RC& c, FMem<MMAX, NS>& M, double* He) {
    const unsigned long long want = __ballot(L.lane < L.N && code_of(L, F_HWP) == 3);
    unsigned long long diff = want ^ L.kmask;
    while (diff) {
        const int k = __ffsll((long long)diff) - 1;
        diff &= diff - 1;
        he_rank1(L, c, M, He, k, ((want >> k) & 1ull) != 0ull);
    }
    L.kmask = want;
    // bilinear terms of the stages whose w is off its bound
    if (L.wmask == 0ull && !__any(wpivot_of(L) >= 0 && wpivot_of(L) != F_FBLO)) return;
    const Tup XF = xi_expr<NS>(L, c, M.ba, GF);
    const Tup W = w_expr<NS>(L, c, M.ba, XF);
    const bool wantP = wpivot_of(L) >= 0 && wpivot_of(L) != F_FBLO;     // on its upper bound w is a constant: nothing to fold
    const bool folded = ((L.wmask >> L.lane) & 1ull) != 0ull;
    const bool same = wantP && folded && W.al == L.fal && W.be == L.fbe && W.ga == L.fga && W.de == L.fde;
    unsigned long long rem = __ballot(folded && !same), addm = __ballot(wantP && !same);
    while (rem) {
        const int k = __ffsll((long long)rem) - 1;
        rem &= rem - 1;
        if (!he_rank2(L, c, M, He, k, bcast(L.fal, k), bcast(L.fbe, k), bcast(L.fga, k), bcast(L.fde, k), false)) L.unsup = 1;
        L.wmask &= ~(1ull << k);
    }
    while (addm) {
        const int k = __ffsll((long long)addm) - 1;
        addm &= addm - 1;
        if (!he_rank2(L, c, M, He, k, bcast(W.al, k), bcast(W.be, k), bcast(W.ga, k), bcast(W.de, k), true)) L.unsup = 1;
        L.wmask |= (1ull << k);
        if (L.lane == k) { L.fal = W.al; L.fbe = W.be; L.fga = W.ga; L.fde = W.de; }
    }
}

// row/column of entry e of a packed lower triangle (e = r(r+1)/2 + c), one table per workgroup
template <int MMAX>
__device__ __forceinline__ unsigned short* rc_table() {
    __shared__ unsigned short tab[MMAX * (MMAX + 1) / 2];
    return tab;
}
template <int MMAX>
__device__ __forceinline__ void rc_table_init() {
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
// working-set list + effective rows from the codes, S = C He C', inverse Schur block P (packed, in place).
// Same three paths as the ABMPC kernel: bordered update after a plain row was added (fast = 1), rank-one
// downdate after a plain row left (fast = 2), full rebuild otherwise.
struct FastInfo { int fast, m_old, kq, tq, drop_pos; double zz; };

template <int MMAX, int NS>
__device__ __forceinline__ int rebuild_and_factor(Lane& L, const RC& c, FMem<MMAX, NS>& M, double* Hs, const FastInfo& F) {
    const int lane = L.lane, N = L.N;
    he_sync(L, c, M, Hs);
    const int cnt = __popcll(codes_eq1(L.code));
    L.base = (int)(scan_excl((double)cnt) + 0.5);
    const int m = (int)(wave_sum((double)cnt) + 0.5);
    if (m > MMAX) return -2;
    {
        const Locals S = locals_of<NS>(L, c, M.ba);
        int pos = L.base;
#pragma unroll
        for (int t = 0; t < kNumF; ++t) {
            if (code_of(L, t) == 1) {
                const Tup R = eff_row<NS>(L, c, M.ba, S, t);
                M.e_al[pos] = R.al; M.e_be[pos] = R.be; M.e_ga[pos] = R.ga; M.e_de[pos] = R.de;
                M.e_d[pos] = R.c; M.w_k[pos] = lane;
                ++pos;
            }
        }
    }
    WSYNC();
    if (m == 0) return 0;
    const unsigned short* rc = rc_table<MMAX>();
    if (F.fast == 1 && m == F.m_old + 1 && F.m_old > 0) {
        const int pl = __popcll(codes_eq1(L.code) & ((1ull << (2 * F.tq)) - 1ull));
        const int p = bcast_i(L.base + pl, F.kq);
        const double iz = 1.0 / F.zz;
        const int nnz = m * (m + 1) / 2;
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
        return m;
    }
    if (F.fast == 2 && m == F.m_old - 1) {
        const int p = F.drop_pos, mo = F.m_old;
        if (lane < mo) M.wa[lane] = M.P[pidx(lane, p)];
        WSYNC();
        const double ip = 1.0 / M.wa[p];
        const int nnz = m * (m + 1) / 2;
        for (int e0 = 0; e0 < nnz; e0 += 64) {
            const int e = e0 + lane;
            double v = 0.0;
            if (e < nnz) {
                const int code = rc[e], r = code >> 8, cc = code & 255;
                const int i = r < p ? r : r + 1, j = cc < p ? cc : cc + 1;
                v = M.P[pidx(i, j)] - M.wa[i] * M.wa[j] * ip;
            }
            WSYNC();
            if (e < nnz) M.P[e] = v;
            WSYNC();
        }
        return m;
    }
    // S columns, two per pass over He
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
            const double c0 = normal_at(L, kj0, M.e_al[j], M.e_be[j], M.e_ga[j], M.e_de[j], bcast(L.Pi, kj0), bcast(L.Th, kj0));
            const double c1 = normal_at(L, kj1, M.e_al[j1], M.e_be[j1], M.e_ga[j1], M.e_de[j1], bcast(L.Pi, kj1), bcast(L.Th, kj1));
            if (lane < NS) { M.ws[lane] = c0; M.wv[lane] = c1; }
            WSYNC();
            double u0, u1;
            hinv_mul2<NS>(Hs, M.ws, M.wv, N, lane, u0, u1);
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
    int singular = 0;
    if (lane < m) M.sv[lane] = fabs(M.P[pidx(lane, lane)]);
    WSYNC();
    const int nnz = m * (m + 1) / 2;
    for (int k = 0; k < m; ++k) {
        const double d = M.P[pidx(k, k)];
        if (!(d > 1e-12 * M.sv[k])) { singular = 1; break; }
        const double inv = 1.0 / d;
        if (lane < m) M.wa[lane] = M.P[pidx(lane, k)];
        WSYNC();
#pragma unroll 2
        for (int e = lane; e < nnz; e += 64) {
            const int code = rc[e], r = code >> 8, cc = code & 255;
            const double c0 = M.wa[r];
            const double cl = M.wa[cc] * inv;
            double v0 = M.P[e] - c0 * cl;
            if (cc == k) v0 = c0 * inv;
            if (r == k) v0 = (cc == k) ? -inv : cl;
            M.P[e] = v0;
        }
        WSYNC();
    }
    if (singular) return -1;
    if (lane < m)
        for (int r = lane; r < m; ++r) M.P[pidx(r, lane)] = -M.P[pidx(r, lane)];
    WSYNC();
    return m;
}

// weights of the objective's gradient on (s_k, v_k, u_k, u_{k-1}) from the local variables that are off their
// bounds at this lane: w_g * xi_g(x), the xi_h penalty, the bilinear price of w
template <int NS>
__device__ __forceinline__ void local_gradient(const Lane& L, const RC& c, const double* ba, double& s, double& v, double& a0, double& a1) {
    s = v = a0 = a1 = 0.0;
    if (L.lane >= L.N) return;
    const Locals S = locals_of<NS>(L, c, ba);
    if (pivot_of(L, GF) >= 0) { s += c.wF * S.XF.al; v += c.wF * S.XF.be; a0 += c.wF * S.XF.ga; a1 += c.wF * S.XF.de; }
    if (pivot_of(L, GS) >= 0) { s += c.wS * S.XS.al; v += c.wS * S.XS.be; a0 += c.wS * S.XS.ga; a1 += c.wS * S.XS.de; }
    if (pivot_of(L, GV) >= 0) { s += c.wV * S.XV.al; v += c.wV * S.XV.be; a0 += c.wV * S.XV.ga; a1 += c.wV * S.XV.de; }
    if (code_of(L, F_HWP) == 3) {
        const double wl = c.wH - c.qH * ba_of<NS>(ba, F_HWP, L.lane);
        s += wl; v += wl * L.chw;
    }
    if (wpivot_of(L) >= 0) {
        const double pr = c.c5 * L.vbar + c.c2;
        s += pr * S.W.al; v += pr * S.W.be + c.c5 * S.W.c; a0 += pr * S.W.ga; a1 += pr * S.W.de;
    }
}

struct LG { double s, v, a0, a1; };     // local_gradient() of the current state (computed once per pass)

// gradient-side vector  [g_eff] + lam_q c_q + C' vec   per lane.  lg != NULL: include g0 and the local-variable terms.
template <int MMAX, int NS>
__device__ __forceinline__ double gradient_side(const Lane& L, const RC& c, FMem<MMAX, NS>& M, int m, const LG* lg, const double* vec,
                                                double lam_q, int kq, double qal, double qbe, double qga, double qde) {
    const int lane = L.lane, N = L.N;
    const bool base = lg != nullptr;
    double s = 0.0, v = 0.0, a0 = 0.0, a1 = 0.0;
    if (base) { s = lg->s; v = lg->v; a0 = lg->a0; a1 = lg->a1; }
    if (lane <= N) { M.ws[lane] = s; M.wv[lane] = v; M.wa[lane] = a0; }
    WSYNC();
    if (lane > 0 && lane < N && a1 != 0.0) atomicAdd(&M.wa[lane - 1], a1);
    if (vec && lane < m) {
        const int ki = M.w_k[lane];
        const double l = vec[lane];
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
    const double g = adjoint<MMAX, NS>(L, M, M.ws, M.wv, M.wa, M.ub);
    return (lane < N) ? g + (base ? L.g0 : 0.0) : 0.0;
}

template <int MMAX, int NS>
__device__ __forceinline__ double rows_dot_img(const FMem<MMAX, NS>& M, int i, int N, const double* x, const double* sx, const double* vx) {
    const int ki = M.w_k[i];
    double s = M.e_al[i] * sx[ki] + M.e_be[i] * vx[ki];
    if (ki < N) s += M.e_ga[i] * x[ki];
    if (ki > 0) s += M.e_de[i] * x[ki - 1];
    return s;
}

template <int MMAX, int NS>
__device__ __forceinline__ void solve_multipliers(FMem<MMAX, NS>& M, int m, int lane, int N) {
    if (lane < m) M.sv[lane] = M.e_d[lane] + rows_dot_img(M, lane, N, M.ub, M.sub, M.vub);
    WSYNC();
    if (lane < m) {
        double acc = 0.0;
        for (int j = 0; j < m; ++j) acc = fma(M.P[pidx(lane, j)], M.sv[j], acc);
        M.lam[lane] = -acc;
    }
    WSYNC();
}

struct Incoming { int kq, qcode, tq, gq; bool is_bound; double al, be, ga, de, d; };

template <int MMAX, int NS>
__device__ __forceinline__ void primal_from_multipliers(Lane& L, const RC& c, FMem<MMAX, NS>& M, const double* Hs, int m, const LG& lg,
                                                        double lam_q, const Incoming& q, double& grad_total) {
    const double g = gradient_side(L, c, M, m, &lg, M.lam, lam_q, q.kq, q.al, q.be, q.ga, q.de);
    grad_total = g;
    if (L.lane < NS) M.ws[L.lane] = g;
    WSYNC();
    L.u = -hinv_mul<NS>(Hs, M.ws, L.N, L.lane);
    hom_traj(L, L.u, L.sh, L.vh);
    L.um1 = lane_prev(L.u);
    if (L.lane < L.N) M.av[L.lane] = L.u;
    if (L.lane <= L.N) { M.shv[L.lane] = L.sh; M.vhv[L.lane] = L.vh; }
    WSYNC();
}

template <int MMAX, int NS>
__device__ __forceinline__ double refine_primal(Lane& L, const RC& c, FMem<MMAX, NS>& M, const double* Hs, int m, const LG& lg,
                                                double lam_q, const Incoming& q, double& grad_total, int max_rounds, double res_tol) {
    primal_from_multipliers(L, c, M, Hs, m, lg, lam_q, q, grad_total);
    if (m == 0) return 0.0;
    double rel0 = 0.0;
    for (int round = 0; round < max_rounds; ++round) {
        double res = 0.0, rel = 0.0;
        if (L.lane < m) {
            res = rows_dot_img(M, L.lane, L.N, M.av, M.shv, M.vhv) - M.e_d[L.lane];
            rel = fabs(res) / (1.0 + fabs(M.e_d[L.lane]));
            M.sv[L.lane] = res;
        }
        int dummy = L.lane;
        wave_argmax(rel, dummy);
        if (round == 0) rel0 = rel;
        if (!(rel > res_tol)) break;
        WSYNC();
        if (L.lane < m) {
            double acc = 0.0;
            for (int j = 0; j < m; ++j) acc = fma(M.P[pidx(L.lane, j)], M.sv[j], acc);
            M.lam[L.lane] += acc;
        }
        WSYNC();
        primal_from_multipliers(L, c, M, Hs, m, lg, lam_q, q, grad_total);
    }
    return rel0;
}

// ----------------------------------------------------------------------------------------------
// multipliers of the pivots and of the bounds of this lane's local variables, as value and rate along the dual
// step.  In: sums over the lane's working-set rows (and the incoming constraint) per group: L = sum lambda,
// R = sum of rates; W sums are weighted with the w-coefficients aw.
struct LocalSums { double LF, LS, LV, LW, RF, RS, RV, RW; };
struct LocalMults { double vF, vS, vV, vW, rF, rS, rV, rW; };      // margin (value, rate): must stay >= 0

__device__ __forceinline__ LocalMults local_mults(const Lane& L, const RC& c, const LocalSums& S, double price, double price_rate) {
    LocalMults o;
    const int pF = pivot_of(L, GF), pW = wpivot_of(L);
    double swv = price + S.LW, swr = price_rate + S.RW;
    o.vS = c.wS - S.LS; o.rS = -S.RS;
    o.vV = c.wV - S.LV; o.rV = -S.RV;
    double vF = c.wF - S.LF, rF = -S.RF;
    const double awF = pF >= 0 ? row_aw(pF, c) : 0.0;
    if (awF != 0.0) {            // pivot of xi_f contains w (w is on its bound then): its multiplier feeds mu_w
        swv += awF * vF; swr += awF * rF;
    }
    if (pW >= 0) {
        const double ia = -1.0 / row_aw(pW, c);
        swv *= ia; swr *= ia;    // multiplier of the pivot row of w
        if (wpivot_in_F(pW)) { vF -= swv; rF -= swr; }
    }
    o.vF = vF; o.rF = rF; o.vW = swv; o.rW = swr;
    return o;
}

// does w matter at this lane (otherwise its price is not watched: w = 0 costs nothing to keep)
__device__ __forceinline__ bool w_relevant(const Lane& L, const RC& c, const Incoming* q, bool have_q) {
    if (L.lane >= L.N) return false;
    if (wpivot_of(L) >= 0) return true;
    bool r = ((codes_eq1(L.code) | codes_eq2(L.code)) & type_range(F_FMLO, F_RTHI)) != 0ull;
    if (have_q && q->kq == L.lane && !q->is_bound && is_wrow(q->tq)) r = true;
    return r;
}

// repair of dual infeasibilities (warm start; also after the one discontinuous event, EV_CAPIN).
// returns 0: nothing to repair, 1: working set changed, 2: only a plain row left (downdate possible)
template <int MMAX, int NS>
__device__ __forceinline__ int warm_repair(Lane& L, const RC& c, const FMem<MMAX, NS>& M, int m, bool single, int& drop_pos) {
    const int lane = L.lane, N = L.N;
    double lmax = 0.0;
    if (lane < m) lmax = fabs(M.lam[lane]);
    lmax = wave_max(lmax);
    const double tol = kTolDual * (1.0 + lmax);
    int changed = 0;
    LocalSums S{0, 0, 0, 0, 0, 0, 0, 0};
    double bestF = -1e300, bestF0 = -1e300, bestS = -1e300, bestV = -1e300, bestW = -1e300;
    int bF = -1, bF0 = -1, bS = -1, bV = -1, bW = -1;
    double worst = tol; int fix = 0x7fffffff;
    int pos = L.base;
#pragma unroll
    for (int t = 0; t < kNumF; ++t) {
        if (code_of(L, t) != 1) continue;
        const int g2 = lane_group(L, t);
        double l = M.lam[pos++];
        if (g2 == GH) {
            if (-l > tol) {
                if (!single) { set_code(L, t, 0); changed = 1; } else if (-l > worst) { worst = -l; fix = (EV_DROP << 16) | (lane << 5) | t; }
            } else if (l - c.wH > tol) {
                if (!single) { set_code(L, t, 3); changed = 1; } else if (l - c.wH > worst) { worst = l - c.wH; fix = (EV_COMPL << 16) | (lane << 5) | t; }
            }
        } else {
            if (-l > tol) {
                if (!single) { set_code(L, t, 0); changed = 1; l = 0.0; } else if (-l > worst) { worst = -l; fix = (EV_DROP << 16) | (lane << 5) | t; }
            }
            const double aw = (lane < N) ? row_aw(t, c) : 0.0;
            if (g2 == GF) { S.LF += l; if (l > bestF) { bestF = l; bF = t; } if (aw == 0.0 && l > bestF0) { bestF0 = l; bF0 = t; } }
            else if (g2 == GS) { S.LS += l; if (l > bestS) { bestS = l; bS = t; } }
            else if (g2 == GV) { S.LV += l; if (l > bestV) { bestV = l; bV = t; } }
            if (aw != 0.0) { S.LW += aw * l; if (is_relax(t) && -aw * l > bestW) { bestW = -aw * l; bW = t; } }
        }
    }
    int capg = 0;
    if (lane < N) {
        const bool wrel = w_relevant(L, c, nullptr, false);
        const LocalMults Mu = local_mults(L, c, S, c.c5 * (L.vbar + L.vh) + c.c2, 0.0);
#pragma unroll
        for (int g2 = GF; g2 <= GW; ++g2) {
            if (g2 == GH) continue;
            if (g2 == GW && !wrel) continue;
            const double val = g2 == GF ? Mu.vF : (g2 == GS ? Mu.vS : (g2 == GV ? Mu.vV : Mu.vW));
            const double sc = g2 == GW ? 1.0 : (1.0 + group_w(c, g2));
            if (-val > tol * sc) {
                if (single) { if (-val / sc > worst) { worst = -val / sc; fix = (EV_CAP << 16) | (lane << 5) | g2; } }
                else if (!changed && !capg) capg = g2;
            }
        }
    }
    int el = -1, et = capg, plain_drop = 0;
    if (single) {
        wave_argmax(worst, fix);
        if (fix != 0x7fffffff) {
            changed = 1;
            const int ek = fix >> 16;
            el = (fix >> 5) & 63; et = fix & 31;
            if (ek == EV_DROP) {
                const int pl = __popcll(codes_eq1(L.code) & ((1ull << (2 * et)) - 1ull));
                drop_pos = bcast_i(L.base + pl, el);
                plain_drop = 1;
            }
            if (lane == el) {
                if (ek == EV_DROP) set_code(L, et, 0);
                else if (ek == EV_COMPL) set_code(L, et, 3);
            }
            if (ek != EV_CAP) el = -1;
        } else el = -1;
    } else if (capg) { el = lane; changed = 1; }
    if (lane == el) {          // margin of a local variable violated: make the member with the largest multiplier its pivot
        if (et == GW) {
            const int p = wpivot_of(L);
            if (p >= 0) set_code(L, p, 0);
            if (bW >= 0 && code_of(L, bW) == 1) set_code(L, bW, 3);
        } else {
            const int p = pivot_of(L, et);
            if (p >= 0) set_code(L, p, 0);
            int bestt = et == GF ? ((wpivot_in_F(wpivot_of(L)) && bF0 >= 0) ? bF0 : bF) : (et == GS ? bS : bV);
            if (bestt >= 0 && code_of(L, bestt) == 1) set_code(L, bestt, 2);
        }
    }
    return __any(changed) ? (plain_drop ? 2 : 1) : 0;
}

// most violated inactive row / bound of a local variable: returns lane*64 + code (code: row type, or 32+group for a
// bound) or -1; best = its scaled violation
template <int NS>
__device__ __forceinline__ int find_violation(const Lane& L, const RC& c, const double* ba, double tolv, double& best) {
    const int lane = L.lane, N = L.N;
    const Locals S = locals_of<NS>(L, c, ba);
    double xiF = 0, xiS = 0, xiV = 0, xiH = 0, wv = 0;
    if (lane < N) { xiF = tup_val(S.XF, L); xiS = tup_val(S.XS, L); xiV = tup_val(S.XV, L); xiH = tup_val(S.XH, L); wv = tup_val(S.W, L); }
    // rows that w relaxes cost nothing to satisfy while the price of w is not positive (predicted speed below zero:
    // a transient of the dual iteration, v_k >= 0 is a hard row)
    const bool wfree = (lane < N) && !(c.c5 * (L.vbar + L.vh) + c.c2 > 0.0);
    // scaled violation val / (1 + |b|): candidates are compared cross-multiplied (no division per row); bestv / bests is
    // this lane's best so far, starting at the tolerance
    double bestv = tolv, bests = 1.0; int myp = -1;
#pragma unroll
    for (int t = 0; t < kNumF; ++t) {
        if (!((L.valid >> t) & 1u) || ((L.ign >> t) & 1u)) continue;
        if (code_of(L, t) != 0) continue;
        if (is_relax(t) && wfree) continue;
        const int g2 = lane_group(L, t);
        const double bt = ba_of<NS>(ba, t, lane);
        double val = row_al(t) * L.sh + row_be(t, c, L.chw) * L.vh + row_ga(t, c) * L.u + row_de(t, lane) * L.um1 - bt;
        if (lane < N) val += row_aw(t, c) * wv;
        val -= (g2 == GF) ? xiF : (g2 == GS ? xiS : (g2 == GV ? xiV : (g2 == GH ? xiH : 0.0)));
        const double sc = 1.0 + fabs(bt);
        if (val * bests > bestv * sc) { bestv = val; bests = sc; myp = t; }
    }
    double myb = bestv / bests;
    if (lane < N) {
        if (pivot_of(L, GF) >= 0 && !((L.ign >> (25 + GF)) & 1u) && L.lbF - xiF > myb) { myb = L.lbF - xiF; myp = 32 + GF; }
        if (pivot_of(L, GS) >= 0 && !((L.ign >> (25 + GS)) & 1u) && L.lbS - xiS > myb) { myb = L.lbS - xiS; myp = 32 + GS; }
        if (pivot_of(L, GV) >= 0 && !((L.ign >> (25 + GV)) & 1u) && L.lbV - xiV > myb) { myb = L.lbV - xiV; myp = 32 + GV; }
        if (code_of(L, F_HWP) == 3 && !((L.ign >> (25 + GH)) & 1u) && L.lbH - xiH > myb) { myb = L.lbH - xiH; myp = 32 + GH; }
        if (wpivot_of(L) >= 0 && !((L.ign >> (25 + GW)) & 1u) && -wv * 1e-3 > myb) { myb = -wv * 1e-3; myp = 32 + GW; }
    }
    best = myb;
    int bp = (myp < 0) ? 0x7fffffff : (lane * 64 + myp);
    wave_argmax(best, bp);
    return bp == 0x7fffffff ? -1 : bp;
}

// effective row of the incoming constraint for the current local states (lane kq computes, everyone receives)
template <int NS>
__device__ __forceinline__ void incoming_row(const Lane& L, const RC& c, const double* ba, Incoming& q) {
    Tup R = tup0();
    if (L.lane == q.kq) {
        const Locals S = locals_of<NS>(L, c, ba);
        if (!q.is_bound) {
            int tq = q.tq;
            // eff_row with a run-time type: select through the unrolled catalogue
#pragma unroll
            for (int t = 0; t < kNumF; ++t) if (t == tq) R = eff_row<NS>(L, c, ba, S, t);
        } else if (q.gq == GW) {
            R.al = -S.W.al; R.be = -S.W.be; R.ga = -S.W.ga; R.de = -S.W.de; R.c = S.W.c;
        } else {
            const Tup& X = xi_of(S, q.gq);
            R.al = -X.al; R.be = -X.be; R.ga = -X.ga; R.de = -X.de; R.c = X.c - group_lb(L, q.gq);
        }
    }
    q.al = bcast(R.al, q.kq); q.be = bcast(R.be, q.kq); q.ga = bcast(R.ga, q.kq); q.de = bcast(R.de, q.kq);
    q.d = bcast(R.c, q.kq);
}

// ----------------------------------------------------------------------------------------------
// the dual active-set solve (structure of the ABMPC kernel's solve_qp, extended by the local variable w)
template <int MMAX, int NS>
__device__ __forceinline__ SolveStats solve_qp(Lane& L, const RC& c, FMem<MMAX, NS>& M, double* Hs, const double* Hbase,
                                               int max_iter, double& grad_total) {
    const int lane = L.lane, N = L.N;
    SolveStats st{0, 0, 0, 0};
    int m = 0;
    bool warm = true, have_q = false;
    int pass = 0;
    double lam_q = 0.0, best = 0.0;
    Incoming q{0, 0, 0, 0, false, 0, 0, 0, 0, 0};
    FastInfo F{0, 0, 0, 0, 0, 1.0};
    int fast_run = 0;
    bool p_stale = false;
    FT_DECL;
    for (;;) {
        if (F.fast != 0 && (++fast_run > 6 || p_stale)) F.fast = 0;
        p_stale = false;
        if (F.fast == 0) fast_run = 0;
        F.m_old = m;
        m = rebuild_and_factor(L, c, M, Hs, F);
        F.fast = 0;
        FT_TOC(4);
        if (__any(L.unsup)) { st.status = 4; break; }
        if (m < 0) {
            if (!warm) { st.status = 2; break; }
            // unusable warm start: cold start from the base inverse
            L.code = 0ull;
            for (int e = lane; e < NS * NS; e += 64) Hs[e] = Hbase[e];
            L.kmask = 0ull; L.wmask = 0ull;
            WSYNC();
            warm = false;
            continue;
        }
        if (have_q) incoming_row<NS>(L, c, M.ba, q);
        LG lg;
        local_gradient<NS>(L, c, M.ba, lg.s, lg.v, lg.a0, lg.a1);
        if (m > 0) {
            const double g = gradient_side(L, c, M, 0, &lg, nullptr, lam_q, q.kq, q.al, q.be, q.ga, q.de);
            if (lane < NS) M.ws[lane] = g;
            WSYNC();
            const double h = hinv_mul<NS>(Hs, M.ws, N, lane);
            double shh, vhh;
            hom_traj(L, h, shh, vhh);
            if (lane < N) M.ub[lane] = h;
            if (lane <= N) { M.sub[lane] = shh; M.vub[lane] = vhh; }
            WSYNC();
            solve_multipliers(M, m, lane, N);
        }
        const double rel0 = refine_primal(L, c, M, Hs, m, lg, lam_q, q, grad_total, 3, 1e-11);
        if (fast_run > 0 && rel0 > 1e-10) p_stale = true;
        FT_TOC(5);
        if (warm) {
            const int rep = warm_repair(L, c, M, m, pass < kSinglePasses, F.drop_pos);
#ifdef EEPACC_FBS_DEBUG
            if (lane == 0 && rep) printf("  repair pass %d rep %d m %d\n", pass, rep, m);
#endif
            if (rep == 2) F.fast = 2;
            if (rep) {
                if (++pass >= kSinglePasses + 6) {
                    L.code = 0ull;
                    for (int e = lane; e < NS * NS; e += 64) Hs[e] = Hbase[e];
                    L.kmask = 0ull; L.wmask = 0ull;
                    WSYNC();
                    warm = false;
                }
                continue;
            }
            warm = false;
        }
        if (!have_q) {
            const int relax_every = 3 * N + 30;
            const double tolv = kTolViol * (st.iters < relax_every ? 1.0 : (st.iters < 2 * relax_every ? 10.0 : (st.iters < 3 * relax_every ? 100.0 : 1000.0)));
            const int bp = find_violation<NS>(L, c, M.ba, tolv, best);
            FT_TOC(6);
            if (bp < 0) break;
            if (++st.iters > max_iter) { st.status = 2; break; }
            q.kq = bp >> 6; q.qcode = bp & 63;
            q.is_bound = q.qcode >= 32;
            q.tq = q.is_bound ? 0 : q.qcode;
            q.gq = q.is_bound ? (q.qcode - 32) : ((q.kq == N) ? GN : group_of(q.qcode));
            have_q = true; lam_q = 0.0;
            incoming_row<NS>(L, c, M.ba, q);
        }
        if (++st.events > 4 * max_iter) { st.status = 2; break; }
        const int kq = q.kq;
        double viol = q.al * M.shv[kq] + q.be * M.vhv[kq] - q.d;
        if (kq < N) viol += q.ga * M.av[kq];
        if (kq > 0) viol += q.de * M.av[kq - 1];
        // u = He c_q and its trajectories
        const double cj = normal_at(L, kq, q.al, q.be, q.ga, q.de, bcast(L.Pi, kq), bcast(L.Th, kq));
        if (lane < NS) M.ws[lane] = cj;
        WSYNC();
        const double ud = hinv_mul<NS>(Hs, M.ws, N, lane);
        double su, vu;
        hom_traj(L, ud, su, vu);
        if (lane < N) M.ub[lane] = ud;
        if (lane <= N) { M.sub[lane] = su; M.vub[lane] = vu; }
        WSYNC();
        double cu = q.al * M.sub[kq] + q.be * M.vub[kq];
        if (kq < N) cu += q.ga * M.ub[kq];
        if (kq > 0) cu += q.de * M.ub[kq - 1];
        double sr = 0.0;
        if (m > 0) {
            if (lane < m) M.sv[lane] = rows_dot_img(M, lane, N, M.ub, M.sub, M.vub);
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
        // rate of the predicted speed along the step (price of w): v_k(t) = v_k - t (sigma_k' z), z = He (c_q - C' r);
        // only needed where w matters
        const bool wrel = w_relevant(L, c, &q, true);
        double vrate = 0.0;
        if (__any(wrel)) {
            double vz = vu;
            if (m > 0) {
                const double gz = gradient_side(L, c, M, m, nullptr, M.rv, 0.0, 0, 0.0, 0.0, 0.0, 0.0);
                if (lane < NS) M.ws[lane] = gz;
                WSYNC();
                const double hz = hinv_mul<NS>(Hs, M.ws, N, lane);
                double sz, vzz;
                hom_traj(L, hz, sz, vzz);
                vz = vu - vzz;
                WSYNC();
            }
            vrate = -vz;
        }
        // blocking events per (lane, type / local variable)
        double t1 = kInf; int ev = 0x7fffffff;
        {
            // candidate step lengths num / den (den > 0) are compared cross-multiplied against the lane's best so far
            // (t1n / t1d): one division per lane at the end instead of one per candidate
            double t1n = kInf, t1d = 1.0;
#define EEPACC_CAND(NUM, DEN, EVCODE) do { const double _n = (NUM), _d = (DEN); \
                if (_n * t1d < t1n * _d) { t1n = _n; t1d = _d; ev = (EVCODE); } } while (0)
            LocalSums S{0, 0, 0, 0, 0, 0, 0, 0};
            int pos = L.base;
#pragma unroll
            for (int t = 0; t < kNumF; ++t) {
                if (code_of(L, t) != 1) continue;
                const int g2 = lane_group(L, t);
                const double l = M.lam[pos], r = M.rv[pos];
                ++pos;
                if (g2 == GH) {
                    if (r > 0.0) EEPACC_CAND(fmax(l, 0.0), r, (EV_DROP << 16) | (lane << 5) | t);
                    else if (r < 0.0) EEPACC_CAND(fmax(c.wH - l, 0.0), -r, (EV_COMPL << 16) | (lane << 5) | t);
                } else {
                    if (r > 0.0) EEPACC_CAND(fmax(l, 0.0), r, (EV_DROP << 16) | (lane << 5) | t);
                    if (g2 == GF) { S.LF += l; S.RF -= r; } else if (g2 == GS) { S.LS += l; S.RS -= r; } else if (g2 == GV) { S.LV += l; S.RV -= r; }
                    if (lane < N) { const double aw = row_aw(t, c); S.LW += aw * l; S.RW -= aw * r; }
                }
            }
            // sign convention: the R sums hold d(lambda)/dt (= -r for working-set rows, +1 for the incoming one)
            if (lane == kq) {
                if (!q.is_bound) {
                    if (q.gq == GF) { S.LF += lam_q; S.RF += 1.0; } else if (q.gq == GS) { S.LS += lam_q; S.RS += 1.0; } else if (q.gq == GV) { S.LV += lam_q; S.RV += 1.0; }
                    if (lane < N) {
                        double awq = 0.0;
#pragma unroll
                        for (int t = F_FMLO; t <= F_RTHI; ++t) if (t == q.tq) awq = row_aw(t, c);
                        S.LW += awq * lam_q; S.RW += awq;
                    }
                } else if (q.gq == GF) { S.LF += lam_q; S.RF += 1.0; }
                else if (q.gq == GS) { S.LS += lam_q; S.RS += 1.0; }
                else if (q.gq == GV) { S.LV += lam_q; S.RV += 1.0; }
                else if (q.gq == GW) { S.LW -= lam_q; S.RW -= 1.0; }
            }
            if (lane < N) {
                const LocalMults Mu = local_mults(L, c, S, c.c5 * (L.vbar + L.vh) + c.c2, c.c5 * vrate);
#pragma unroll
                for (int g2 = GF; g2 <= GW; ++g2) {
                    if (g2 == GH) continue;
                    if (g2 == GW) {
                        if (!wrel) continue;
                        // on its bound w can only leave through a relaxing row that is active (or coming in)
                        if (wpivot_of(L) < 0) {
                            bool cand = code_of(L, F_FMLO) == 1 || code_of(L, F_TQMIN) == 1 || code_of(L, F_RTLO) == 1;
                            if (lane == kq && !q.is_bound && is_relax(q.tq)) cand = true;
                            if (!cand) continue;
                        }
                    }
                    const double val = g2 == GF ? Mu.vF : (g2 == GS ? Mu.vS : (g2 == GV ? Mu.vV : Mu.vW));
                    const double rate = g2 == GF ? Mu.rF : (g2 == GS ? Mu.rS : (g2 == GV ? Mu.rV : Mu.rW));
                    if (rate < 0.0) {
                        EEPACC_CAND(fmax(val, 0.0), -rate, (EV_CAP << 16) | (lane << 5) | g2);
                    }
                }
            }
            if (lane == kq && !q.is_bound && q.gq == GH) {
                EEPACC_CAND(fmax(c.wH - lam_q, 0.0), 1.0, (EV_CAPIN << 16) | (lane << 5));
            }
            if (lane == kq && q.is_bound && q.gq == GH) {
                const double xi_now = L.lbH - viol, den = 1.0 - c.qH * zz;
                if (den > 0.0) {
                    EEPACC_CAND(fmax(c.wH + c.qH * xi_now - lam_q, 0.0), den, (EV_DROPH << 16) | (lane << 5) | F_HWP);
                }
            }
#undef EEPACC_CAND
            t1 = (ev == 0x7fffffff) ? kInf : t1n / t1d;
            wave_argmin(t1, ev);
        }
        const double tstep = fmin(t1, t2);
        if (!(tstep < 1e299)) {
            if (best < 1e-7) {
                if (lane == kq) L.ign |= q.is_bound ? (1u << (25 + q.gq)) : (1u << q.qcode);
                have_q = false; lam_q = 0.0; continue;
            }
            st.status = 1; break;
        }
        lam_q += tstep;
#ifdef EEPACC_FBS_DEBUG
        if (lane == 0) printf("  it %d q k=%d code=%d viol %.3e t %.3e %s ev=%d lane=%d t=%d m %d zz %.3e\n", st.iters, kq, q.qcode, viol, tstep,
                              t2 <= t1 ? "full" : "event", ev >> 16, (ev >> 5) & 63, ev & 31, m, zz);
#endif
        bool finished = false;
        if (t2 <= t1) {
            // full step: the incoming constraint becomes active
            // w reaches its upper bound (Fb = -1e4 N) while a row defines it: the bound row takes over as w's pivot (the
            // canonical form of "w on its upper bound": free of xi_f, so rows that contain w may define xi_f) and the
            // former pivot becomes an ordinary working-set row
            const bool w_upper = !q.is_bound && q.tq == F_FBLO && bcast_i(wpivot_of(L), kq) >= 0;
            if (!q.is_bound && m > 0 && !w_upper) { F.fast = 1; F.kq = kq; F.tq = q.tq; F.zz = zz; }
            if (lane == kq) {
                if (w_upper) { set_code(L, wpivot_of(L), 1); set_code(L, F_FBLO, 3); }
                else if (!q.is_bound) set_code(L, q.tq, 1);
                else if (q.gq == GH) set_code(L, F_HWP, 1);
                else if (q.gq == GW) { const int p = wpivot_of(L); set_code(L, p, 1); }
                else { const int p = pivot_of(L, q.gq); set_code(L, p, 1); }
            }
            finished = true;
        } else {
            const int ek = ev >> 16, el = (ev >> 5) & 63, et = ev & 31;
            if (ek == EV_DROP) {
                const int pl = __popcll(codes_eq1(L.code) & ((1ull << (2 * et)) - 1ull));
                F.drop_pos = bcast_i(L.base + pl, el);
                F.fast = 2;
                if (lane == el) set_code(L, et, 0);
            }
            else if (ek == EV_COMPL) { if (lane == el) set_code(L, et, 3); }
            else if (ek == EV_DROPH) { if (lane == el) set_code(L, et, 0); finished = true; }
            else if (ek == EV_CAPIN) {
                // the slack jumps off its bound by the remaining violation: the only discontinuous event; the
                // multipliers of the working set are re-validated (repair pass) before the next constraint
                if (lane == el) set_code(L, F_HWP, 3);
                finished = true; warm = true; pass = 0;
            }
            else if (ek == EV_CAP) {
                int fin = 0;
                if (lane == el) {
                    const int g2 = et;
                    // candidates: active members of the group with their (weighted) multipliers after the step
                    int bestm = -1, bestm0 = -1; double bl = -1e300, bl0 = -1e300;
                    bool fmlo_cand = false;
                    int pos = L.base;
#pragma unroll
                    for (int t = 0; t < kNumF; ++t) {
                        if (code_of(L, t) != 1) continue;
                        const double l = M.lam[pos] - tstep * M.rv[pos];
                        ++pos;
                        const double aw = row_aw(t, c);
                        if (g2 == GW) { if (is_relax(t) && -aw * l > bl) { bl = -aw * l; bestm = t; } if (t == F_FMLO) fmlo_cand = true; }
                        else if (t >= F_TQMIN && t <= F_VINC && group_of(t) == g2) {
                            if (l > bl) { bl = l; bestm = t; }
                            if (aw == 0.0 && l > bl0) { bl0 = l; bestm0 = t; }
                        }
                    }
                    const bool q_here = (kq == el) && !q.is_bound;
                    if (g2 == GW) {
                        const int p = wpivot_of(L);
                        const int pF = pivot_of(L, GF);
                        bool q_row_here = q_here && is_relax(q.tq);
                        if (pF >= 0 && row_aw(pF, c) != 0.0) {
                            // xi_f is defined through a row that contains w: w may only be pinned by a row free of xi_f
                            bestm = fmlo_cand ? (int)F_FMLO : -1;
                            q_row_here = q_row_here && q.tq == F_FMLO;
                            if (bestm < 0 && !q_row_here && bl > -1e299) L.unsup = 1;      // (xi_f, w) coupled through their pivots
                        }
                        const bool q_bound_here = (kq == el) && q.is_bound && q.gq == GW;
                        if (p >= 0) set_code(L, p, 0);
                        if (bestm >= 0) set_code(L, bestm, 3);
                        else if (q_row_here) { set_code(L, q.tq, 3); fin = 1; }
                        else if (q_bound_here) fin = 1;
                        // else: the price of w vanished (v_k fell to zero along the path): w is free, its pivot row is let go
                    } else {
                        const int p = pivot_of(L, g2);
                        const bool q_row_here = q_here && q.gq == g2;
                        const bool q_bound_here = (kq == el) && q.is_bound && q.gq == g2;
                        // a pivot of xi_f that contains w while w is off its bound would couple the two eliminations
                        if (g2 == GF && wpivot_in_F(wpivot_of(L))) { bestm = bestm0; if (bestm0 < 0 && bl > -1e299) L.unsup = 1; }
                        if (p < 0) {                       // on the bound -> off it
                            if (bestm < 0) { if (q_row_here) { set_code(L, q.tq, 2); fin = 1; } else L.unsup = 1; }
                            else set_code(L, bestm, 2);
                        } else {                           // pivot multiplier reached zero
                            set_code(L, p, 0);
                            if (bestm >= 0) set_code(L, bestm, 2);
                            else if (q_row_here) { set_code(L, q.tq, 2); fin = 1; }
                            else if (q_bound_here) fin = 1;
                            else L.unsup = 1;              // driven to zero by the pivot row of w: coupled state
                        }
                    }
                }
                if (__any(fin)) finished = true;
            }
        }
        if (finished) { have_q = false; lam_q = 0.0; q.al = q.be = q.ga = q.de = q.d = 0.0; }
        FT_TOC(7);
    }
    if (st.status == 0 && m > 0) {
        LG lg;
        local_gradient<NS>(L, c, M.ba, lg.s, lg.v, lg.a0, lg.a1);
        refine_primal(L, c, M, Hs, m, lg, 0.0, q, grad_total, 4, 1e-14);
    }
    FT_TOC(8);
    FT_FLUSH();
    st.m = m;
    return st;
}

// ----------------------------------------------------------------------------------------------
// per-step set-up and extraction (F4)

struct StepIn { double s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, v_prev; int k_step, have_vprev; };
struct StepOut { double out[EEPACC_OUT_N]; int status, iters; };

// LDS layout of a block: per wave [FMem][He NS x NS]; the step's base inverse (needed again only by a cold restart)
// is kept in a per-wave global scratch
__host__ __device__ inline size_t wave_bytes(size_t wm, int ns) {
    return ((wm + (size_t)ns * ns * sizeof(double)) + 15) & ~(size_t)15;
}

// One FBMPC step of the wave's instance (ABO/RunOpt_FBMPC.m:204-320).  A22, D2: this lane's carried state-space
// entries (the A(k)/D(k) index quirk :247-259 freezes stage k at MPC step k); code: warm working set.
template <int MMAX, int NS>
__device__ __forceinline__ void fb_step(const DevCfg& C, FMem<MMAX, NS>& M, double* Hs, double* Hb, const StepIn& in,
                                        double& A22, double& D2, unsigned long long& code, StepOut& so,
                                        double& s_pred, double& v_pred, double ps_prev, double pv_prev) {
    Lane L;
    L.lane = lane_id(); L.N = C.N;
    const int lane = L.lane, N = C.N;
    const double lm = C.lambda * C.m, za = C.zeta_a;
    L.T = lane < N ? C.Tvec[lane] : 0.0;
    FT_DECL;
    RC c;
    c.tau_min = C.tau_min;
    c.c1 = C.phi * C.T_m_max * C.T_m_max / 4.0 / C.P_m_max;
    c.g_tqmin = -C.eta_TF / C.phi; c.g_tqmax = 1.0 / C.eta_TF / C.phi;
    c.Lmu = C.L / C.mu; c.g_rtlo = -(c.Lmu + C.h_g); c.g_rthi = c.Lmu - C.h_g;
    c.wF = C.fb_w[6]; c.wS = C.fb_w[5]; c.wV = C.fb_w[3]; c.wH = 100.0 * C.fb_w[4]; c.qH = 2.0 * C.fb_w[4];
    const double Kr = (30.0 / 3.14159265358979323846) * C.phi;
    c.c5 = C.fb_w[0] * Kr * C.b_quadr[4]; c.c2 = C.fb_w[0] * C.b_quadr[1];
    // estimators (A2)
    double s_est, v_est, stv_est, vtv_est;
    if (C.paramEstSetting == 2) {
        // EstimateVehicleTrajectory.m:81-88: [x_curr; prev(3:end); prev(end) + Ts v_prev(end)]; ps_prev/pv_prev hold
        // the previous prediction of stage lane+1 (stage N for the last two lanes)
        s_est = lane == 0 ? in.s : (lane < N ? ps_prev : ps_prev + C.Tvec[N - 1] * pv_prev);
        v_est = lane == 0 ? in.v : pv_prev;
    } else {
        estimate_traj(C, C.paramEstSetting, C.tConstACC_ego, in.s, in.v, in.a_prev, lane, s_est, v_est);
    }
    estimate_traj(C, C.TVestSetting, C.tConstACC_tar, in.s_tv, in.v_tv, in.a_tv_prev, lane, stv_est, vtv_est);
    const double dist_hor = bcast(s_est, N) - in.s;                           // :208
    const double stv_Nm1 = bcast(stv_est, N - 1);
    double v_lim, v_curv, v_stop, v_TL, a_min, a_max, j_min, j_max;
    route_bounds(C, s_est, v_est, in.t0, lane < N ? lane : N - 1, v_lim, v_curv, v_stop, v_TL, a_min, a_max, j_min, j_max);
    double sn, cs;
    slope_trig(C, s_est, sn, cs);                                             // theta_est(k) (CreateQP_FB.m:51-52)
    const double zrg = C.m * C.g * (C.c_r * cs + sn);
    const double zw = C.m * C.g * (C.L_f * cs + C.h_g * sn);
    const double zrg_p = lane_prev(zrg), ve_p = lane_prev(v_est);
    const double Dz = lane > 0 ? zrg - zrg_p : 0.0;
    // state-space model with the A(k)/D(k) index quirk (ABO/RunOpt_FBMPC.m:78-90, 247-259)
    if (in.k_step == 0) {
        if (C.FBuseTaylor) { A22 = 1.0 - 2.0 * L.T * za * in.v * in.v / lm; D2 = L.T / lm * (za * in.v * in.v); }
        else { A22 = 1.0; D2 = L.T / lm * (-za * in.v * in.v); }
    }
    if (C.FBuseTaylor) {
        const double vi = bcast(v_est, N - 1), zi = bcast(zrg, N - 1), Ti = C.Tvec[N - 1];
        if (lane == in.k_step && lane < N) { A22 = 1.0 - 2.0 * Ti * za * vi / lm; D2 = Ti / lm * (za * vi * vi - zi); }
    } else if (lane < N) {
        D2 = L.T / lm * (-za * v_est * v_est - zrg);
    }
    const double a22 = lane < N ? A22 : 1.0, d2 = lane < N ? D2 : 0.0;
    // condensing scalings
    L.Pi = scan_prod_excl(a22);
    const double Pi1 = L.Pi * a22;
    L.Th = scan_excl(lane < N ? L.T * L.Pi : 0.0);
    L.Th1 = L.Th + (lane < N ? L.T * L.Pi : 0.0);
    L.gam = lane < N ? (L.T / lm) / Pi1 : 0.0;
    if (lane > N) L.Pi = 0.0;
    // free response: vbar_k = Pi_k (v_0 + sum_{i<k} D2_i / Pi_{i+1}), sbar_k = s_0 + sum_{i<k} T_i vbar_i
    L.vbar = L.Pi * (in.v + scan_excl(lane < N ? d2 / Pi1 : 0.0));
    const double sbar = in.s + scan_excl(lane < N ? L.T * L.vbar : 0.0);
    // sparse-form objective over (v_k, u_k), block tridiagonal (CreateQP_FB.m:181-208)
    const double w_P = C.fb_w[0], w_a = C.fb_w[1], w_j = C.fb_w[2];
    double qvv_d = 0, qvv_o = 0, qa = 0, qb = 0, qc = 0, qd = 0, qe = 0, cvv = 0, cuu = 0;
    {
        const double fj = lane < N ? 2.0 * w_j / ((lm * L.T) * (lm * L.T)) : 0.0;
        const double fj_n = lane_next(fj), ve_n = lane_next(v_est), Dz_n = lane_next(Dz);
        if (lane < N) {
            const double fa = 2.0 * w_a / (lm * lm);
            qvv_d = w_P * 2.0 * Kr * Kr * C.b_quadr[5] + fa * (za * za * v_est * v_est + za * zrg);
            qa = c.c5 + fa * (-za * v_est);
            qd = fa;
            cvv = w_P * Kr * C.b_quadr[2];
            cuu = c.c2 - 2.0 * w_a * zrg / (lm * lm);
            if (lane == 0) {
                qd += fj;
                cuu -= 2.0 * w_j * (za * in.v * in.v + zrg + lm * in.a_prev) / ((lm * L.T) * (lm * L.T));
            } else {
                // lower-right block of this stage's 6x6 jerk coupling and the coupling to stage k-1
                qvv_d += fj * (za * za * ve_p * ve_p - 2.0 * za * Dz);
                qa += fj * (-za * v_est);
                qd += fj;
                qvv_o = fj * (-za * za * v_est * ve_p);
                qb = fj * (za * ve_p);          // v_{k-1} u_k
                qc = fj * (za * v_est);         // v_k u_{k-1}
                qe = -fj;
                cuu -= fj * Dz;
            }
            if (lane + 1 < N) {
                // upper-left block of stage k+1's coupling lands on this stage
                qvv_d += fj_n * (za * za * ve_n * ve_n + 2.0 * za * Dz_n);
                qa += fj_n * (-za * v_est);
                qd += fj_n;
                cuu += fj_n * Dz_n;
            }
        }
    }
    // base gradient: g = cu + Qvu' vbar + Psi_v' (cv + Qvv vbar)
    {
        const double vb_p = lane_prev(L.vbar), vb_n = lane_next(L.vbar);
        const double qvvo_n = lane_next(qvv_o), qc_n = lane_next(qc);
        double gv = 0.0, gu = 0.0;
        if (lane < N) {
            gv = cvv + qvv_d * L.vbar + (lane > 0 ? qvv_o * vb_p : 0.0) + (lane + 1 < N ? qvvo_n * vb_n : 0.0);
            gu = cuu + qa * L.vbar + (lane > 0 ? qb * vb_p : 0.0) + (lane + 1 < N ? qc_n * vb_n : 0.0);
        }
        if (lane <= N) { M.ws[lane] = 0.0; M.wv[lane] = gv; M.wa[lane] = gu; }
        WSYNC();
        L.g0 = adjoint<MMAX, NS>(L, M, M.ws, M.wv, M.wa, M.ub);
    }
    // dense Hessian of the condensed objective, column `lane`, from closed forms (header comment); stage arrays
    // staged through LDS: gam -> ws, R -> wv, pa -> wa, pb -> ub, pc -> sub, exc -> vub
    {
        const double Pi_p = lane_prev(L.Pi);
        const double qvvo_n = lane_next(qvv_o), qc_n = lane_next(qc), qe_n = lane_next(qe);
        const double Pi_n = Pi1;            // Pi_{k+1}
        double rho = 0.0;
        if (lane < N) rho = L.Pi * ((lane > 0 ? qvv_o * Pi_p : 0.0) + qvv_d * L.Pi + (lane + 1 < N ? qvvo_n * Pi_n : 0.0));
        const double inc = scan_incl(rho);
        const double R = read_lane63(inc) - inc;             // sum over stages > lane
        const double exc = (lane + 1 < N) ? Pi_n * qvvo_n * L.Pi : 0.0;
        const double pa = (lane > 0 && lane < N) ? Pi_p * qb : 0.0;
        const double pb = lane < N ? L.Pi * qa : 0.0;
        const double pc = (lane + 1 < N) ? Pi_n * qc_n : 0.0;
        if (lane <= N) { M.ws[lane] = L.gam; M.wv[lane] = R; M.wa[lane] = pa; M.ub[lane] = pb; M.sub[lane] = pc; M.vub[lane] = exc; }
        WSYNC();
        if (lane < NS) {
            const int j = lane;
            for (int i = 0; i < NS; ++i) {
                double h = 0.0;
                if (i < N && j < N) {
                    const double gi = M.ws[i];
                    const double t1 = gi * ((i < j - 1 ? pa : 0.0) + (i < j ? pb : 0.0) + (i < j + 1 ? pc : 0.0));
                    const double t2 = L.gam * ((j < i - 1 ? M.wa[i] : 0.0) + (j < i ? M.ub[i] : 0.0) + (j < i + 1 ? M.sub[i] : 0.0));
                    const double Rm = i > j ? M.wv[i] : R;
                    const double t3 = gi * L.gam * (Rm - (i == j ? exc : 0.0));
                    const double quu = i == j ? qd : (i == j - 1 ? qe : (i == j + 1 ? qe_n : 0.0));
                    h = quu + (t1 + t2) + t3;
                }
                Hs[i * NS + j] = h;
            }
        }
        WSYNC();
    }
#ifdef EEPACC_FBS_DEBUG
    if (lane < N) printf("DBG lane %d g0 %.15e Pi %.15e gam %.15e vbar %.15e H0 %.15e H5 %.15e Hd %.15e A22 %.15e D2 %.15e\n", lane, L.g0, L.Pi, L.gam, L.vbar,
                         Hs[0 * NS + lane], Hs[5 * NS + lane], Hs[lane * NS + lane], A22, D2);
#endif
    FT_TOC(0);
    // in-place inverse by symmetric sweeps (H is positive definite: cond ~ 1e2); afterwards Hs = -H^-1
    int h_bad = 0;
    {
        // all 64 lanes work: lane l updates rows [r0, r0 + RPL) of column l % NS (NS = 32: two lanes per column)
        constexpr int HALVES = 64 / NS, RPL = NS / HALVES;
        const int jcol = lane & (NS - 1), r0 = (lane / NS) * RPL;
        double* col = Hs + jcol;
        for (int k = 0; k < N; ++k) {
            const double d = Hs[k * NS + k];
            if (!(d > 0.0)) { h_bad = 1; break; }
            const double inv = 1.0 / d;
            if (lane < NS) M.wa[lane] = (lane < N) ? Hs[k * NS + lane] : 0.0;
            WSYNC();
            const double hkj = M.wa[jcol];
            const double f = hkj * inv;
            const bool piv = jcol == k;
#pragma unroll
            for (int ii = 0; ii < RPL; ++ii) {
                const int i = r0 + ii;
                const double ck = M.wa[i], old = col[i * NS];
                const double upd = piv ? ck * inv : fma(-ck, f, old);
                col[i * NS] = (i == k) ? (piv ? -inv : f) : upd;
            }
            WSYNC();
        }
    }
    for (int e = lane; e < NS * NS; e += 64) { const double x = -Hs[e]; Hs[e] = x; Hb[e] = x; }
    WSYNC();
    FT_TOC(1);
    // rows: right-hand sides relative to the free response (CreateQP_FB.m:311-489)
    const double T_hwp = 2.0, A_hwp = 2.0, G_hwp = -0.0246 * T_hwp + 0.010819;
    L.chw = C.FBuseTaylor ? T_hwp + 2.0 * G_hwp * v_est : T_hwp + G_hwp * v_est;
    {
        const double base = za * v_est * v_est + zrg;
        double b[kNumF];
        b[F_SLO] = -in.s; b[F_SHI] = C.s_goal; b[F_VLO] = -0.0; b[F_VHI] = C.v_max;
        b[F_FMLO] = 1e4; b[F_FMHI] = 1e4; b[F_FBLO] = 1e4;
        b[F_TQMIN] = C.T_m_max; b[F_TQMAX] = C.T_m_max;
        b[F_RTLO] = zw - C.h_g * zrg; b[F_RTHI] = zw - C.h_g * zrg;
        b[F_FTHI] = C.mu * C.m * C.g * cs; b[F_FTLO] = C.mu * C.m * C.g * cs;
        b[F_AMAX] = lm * a_max + base; b[F_AMIN] = -(lm * a_min + base);
        if (lane == 0) {
            b[F_JMAX] = lm * (L.T * j_max + in.a_prev) + base;
            b[F_JMIN] = -(lm * (L.T * j_min + in.a_prev) + base);
        } else {
            const double dj = za * (v_est * v_est - ve_p * ve_p) + Dz;
            b[F_JMAX] = lm * L.T * j_max + dj;
            b[F_JMIN] = -(lm * L.T * j_min + dj);
        }
        b[F_VLIM] = v_lim; b[F_VCURV] = v_curv; b[F_VSTOP] = v_stop; b[F_VTL] = v_TL;
        b[F_VINC] = -fmin(v_lim, v_curv);
        b[F_SAFE1] = stv_est - C.h_min; b[F_SAFE2] = stv_est;
        b[F_HWP] = C.FBuseTaylor ? stv_est - A_hwp + G_hwp * v_est * v_est : stv_est - A_hwp;
        if (lane == N) { b[F_SAFE1] = stv_Nm1 - C.h_min; b[F_SAFE2] = stv_Nm1; }
        unsigned valid = 0u;
        L.lbF = L.lbS = L.lbV = L.lbH = 0.0;
        int infeasible_const = 0;
#pragma unroll
        for (int t = 0; t < kNumF; ++t) {
            const double al = row_al(t), be = row_be(t, c, L.chw);
            const double bat = b[t] - al * sbar - be * L.vbar;
            if (owns_slot(t) && lane <= N) M.ba[slot_of(t) * (NS + 1) + lane] = bat;
            bool exists;
            if (lane < N) {
                exists = true;
                if (t == F_SHI && !(C.s_goal < 1e300)) exists = false;
            } else exists = (lane == N) && (t == F_SAFE1 || t == F_SAFE2);
            if (exists && lane == 0 && row_ga(t, c) == 0.0 && row_aw(t, c) == 0.0) {
                // stage-0 rows on (s_0, v_0) only are constants: fold into slack bounds
                exists = false;
                const int g2 = group_of(t);
                const double need = -bat;
                if (g2 == GN) { if (need > kTolViol * (1.0 + fabs(b[t]))) infeasible_const = 1; }   // the dense solver's row tolerance
                else if (g2 == GF) L.lbF = fmax(L.lbF, need);
                else if (g2 == GS) L.lbS = fmax(L.lbS, need);
                else if (g2 == GV) L.lbV = fmax(L.lbV, need);
                else L.lbH = fmax(L.lbH, need);
            }
            if (exists) valid |= (1u << t);
        }
        L.valid = valid;
        so.status = __any(infeasible_const) ? 1 : 0;
    }
    // Feasibility of the hard rows, decided before the solve.  Only the bounds on s_k, v_k, the force bounds and the two
    // terminal headway rows have no slack (CreateQP_FB.m:311-326,476-489).  v_{k+1} = A22_k v_k + beta_k u_k + D2_k and
    // s_{k+1} = s_k + T_k v_k are monotone in u, so braking as hard as the force bounds allow without letting v drop
    // below zero gives the smallest reachable s_N and v_N at once: if even that violates a terminal row the QP has no
    // solution (a lead vehicle too close to stop behind).  A dual active set needs its whole iteration budget to find
    // that out; decided here, the step costs nothing, reports status 1 like the reference and applies full braking.
    bool hopeless = false;
    {
        double vmin = in.v, smin = in.s;
        const double umin = -2e4;                         // Fm >= -1e4 and Fb >= -1e4 (b[F_FMLO], b[F_FBLO] above)
        for (int k = 0; k < N; ++k) {
            const double ak = bcast(a22, k), dk = bcast(d2, k), Tk = bcast(L.T, k);
            smin += Tk * vmin;
            vmin = fmax(ak * vmin + dk + Tk / lm * umin, 0.0);
        }
        const double over = fmax(smin - (stv_Nm1 - C.h_min), smin + c.tau_min * vmin - stv_Nm1);
        hopeless = over > 1e-6 * (1.0 + fabs(stv_Nm1));
        if (hopeless) so.status = 1;
    }
    L.ign = 0u;
    L.code = code;
#pragma unroll
    for (int t = 0; t < kNumF; ++t)
        if (!((L.valid >> t) & 1u)) set_code(L, t, 0);
    L.u = L.sh = L.vh = L.um1 = 0.0;
    L.kmask = 0ull; L.wmask = 0ull; L.fal = L.fbe = L.fga = L.fde = 0.0;
    L.unsup = 0; L.base = 0;
    WSYNC();
    double grad_total = 0.0;
    SolveStats st{2, 0, 0, 0};
    FT_TOC(2);
    FT_FLUSH();
#ifdef EEPACC_FBS_TIMING
    for (int _i = 0; _i < 16; ++_i) _fp[_i] = 0;
#endif
    // iteration cap: a cold solve needs about 1.5 N working-set changes; beyond 6 N + 60 the solve is cycling (status 1)
    if (!h_bad && !hopeless) st = solve_qp<MMAX, NS>(L, c, M, Hs, Hb, 6 * N + 60, grad_total);
    code = L.code;
#ifdef EEPACC_FBS_DEBUG
    if (lane <= N) printf("FIN lane %d code %llx u %.15e ign %x valid %x st %d iters %d m %d\n", lane, L.code, L.u, L.ign, L.valid, st.status, st.iters, st.m);
#endif
    // recover z = Psi x + d (F4): predicted states, stage-0 controls and slacks
    s_pred = sbar + L.sh; v_pred = L.vbar + L.vh;
    const Locals S = locals_of<NS>(L, c, M.ba);
    double xiF = 0, xiS = 0, xiV = 0, xiH = 0, w = 0;
    if (lane < N) {
        xiF = fmax(tup_val(S.XF, L), L.lbF); xiS = fmax(tup_val(S.XS, L), L.lbS); xiV = fmax(tup_val(S.XV, L), L.lbV);
        xiH = fmax(tup_val(S.XH, L), L.lbH);
        w = fmax(tup_val(S.W, L), 0.0);
    }
    // sol.cost = f(z) - f(d): stage terms of the sparse-form objective evaluated directly
    double part = 0.0;
    {
        const double vh_p = lane_prev(L.vh), vh_n = lane_next(L.vh), u_n = lane_next(L.u);
        const double qvvo_n = lane_next(qvv_o), qb_n = lane_next(qb), qc_n = lane_next(qc), qe_n = lane_next(qe);
        const double vb_p = lane_prev(L.vbar), vb_n = lane_next(L.vbar);
        if (lane < N) {
            const double gv = cvv + qvv_d * L.vbar + (lane > 0 ? qvv_o * vb_p : 0.0) + (lane + 1 < N ? qvvo_n * vb_n : 0.0);
            const double gu = cuu + qa * L.vbar + (lane > 0 ? qb * vb_p : 0.0) + (lane + 1 < N ? qc_n * vb_n : 0.0);
            const double Qv = qvv_d * L.vh + (lane > 0 ? qvv_o * vh_p : 0.0) + (lane + 1 < N ? qvvo_n * vh_n : 0.0);
            const double Qvu_u = qa * L.u + (lane > 0 ? qc * L.um1 : 0.0) + (lane + 1 < N ? qb_n * u_n : 0.0);    // (Qvu u)_k
            const double Quu_u = qd * L.u + (lane > 0 ? qe * L.um1 : 0.0) + (lane + 1 < N ? qe_n * u_n : 0.0);
            part = L.vh * gv + L.u * gu + 0.5 * L.vh * Qv + L.vh * Qvu_u + 0.5 * L.u * Quu_u;
            part += (c.c5 * (L.vbar + L.vh) + c.c2) * w;
            part += c.wF * xiF + c.wS * xiS + c.wV * xiV + c.wH * xiH + 0.5 * c.qH * xiH * xiH;
        }
    }
    const double cost = wave_sum(part);
    const double u0 = bcast(L.u, 0), w0 = bcast(w, 0);
    so.out[EEPACC_OUT_S] = in.s;
    so.out[EEPACC_OUT_V] = in.v;
    double Fm0 = u0 + w0, Fb0 = -w0;                                            // :294-299
    if (so.status != 0 || st.status != 0) {
        // no solution (or the iteration gave up): the last iterate is applied like the reference does
        // (opts.error_on_fail = false), projected on the actuators' hard bounds (CreateQP_FB.m:319-326) so that the
        // plant state stays finite whatever the iterate was
        Fm0 = (Fm0 == Fm0) ? fmin(fmax(Fm0, -1e4), 1e4) : 0.0;
        Fb0 = (Fb0 == Fb0) ? fmin(fmax(Fb0, -1e4), 0.0) : 0.0;
        if (hopeless) { Fm0 = in.v > 1e-3 ? -1e4 : 0.0; Fb0 = Fm0; }     // no solve was run: brake as hard as the bounds allow
    }
    so.out[EEPACC_OUT_FM] = Fm0;
    so.out[EEPACC_OUT_FB] = Fb0;
    so.out[EEPACC_OUT_A] = in.have_vprev ? (in.v - in.v_prev) / C.Tvec[0] : 0.0;  // :316-318
    so.out[EEPACC_OUT_XI_V] = bcast(xiV, 0);
    so.out[EEPACC_OUT_XI_H] = bcast(xiH, 0);
    so.out[EEPACC_OUT_XI_S] = bcast(xiS, 0);
    so.out[EEPACC_OUT_XI_F] = bcast(xiF, 0);
    so.out[EEPACC_OUT_COST] = cost;
    so.out[EEPACC_OUT_DISTHOR] = dist_hor;
    so.out[EEPACC_OUT_AQP] = 0.0;
    if (st.status != 0) so.status = 1;
    if (so.status != 0) {
        // a failed step may sit on a physically meaningless measured state (negative speed after earlier failures): its
        // slack and cost read-outs carry no information; they are reported as zero rather than as inf / NaN
        for (int f = EEPACC_OUT_XI_V; f <= EEPACC_OUT_COST; ++f) {
            const double x = so.out[f];
            if (!(fabs(x) < 1e300)) so.out[f] = 0.0;
        }
    }
    so.iters = st.iters;
#ifdef EEPACC_FBS_TIMING
    _ft = wall_clock64() - 0;   // output phase is measured from the end of the solve by the caller's next step; negligible
    if (lane == 0) atomicAdd(&g_fbs_prof[15], 1ull);
#endif
}

// receding-horizon shift of the working set: stage k takes stage k+1's codes, the last stage and the terminal rows
// keep theirs
__device__ __forceinline__ unsigned long long shift_codes(unsigned long long code, int N) {
    const int lane = lane_id();
    unsigned lo = (unsigned)code, hi = (unsigned)(code >> 32);
    unsigned nlo = __shfl_down(lo, 1, 64), nhi = __shfl_down(hi, 1, 64);
    unsigned long long nxt = ((unsigned long long)nhi << 32) | nlo;
    if (lane < N - 1) return nxt;
    return code;
}

template <int MMAX, int NS>
__device__ FMem<MMAX, NS>* wave_mem(unsigned char* smem, double*& He) {
    unsigned char* base = smem + wave_bytes(sizeof(FMem<MMAX, NS>), NS) * (threadIdx.x >> 6);
    He = reinterpret_cast<double*>(base + sizeof(FMem<MMAX, NS>));
    return reinterpret_cast<FMem<MMAX, NS>*>(base);
}

// per-instance state block: [codes 64 x u64 | A22 64 | D2 64 | s_pred 64 | v_pred 64]
__device__ __forceinline__ double* st_A22(double* s) { return s + 64; }
__device__ __forceinline__ double* st_D2(double* s) { return s + 128; }
__device__ __forceinline__ double* st_sp(double* s) { return s + 192; }
__device__ __forceinline__ double* st_vp(double* s) { return s + 256; }

__device__ __forceinline__ void write_out(double* dst, size_t stride_field, const StepOut& so, int lane) {
    if (lane < EEPACC_OUT_N) {
        double val = 0.0;
#pragma unroll
        for (int f = 0; f < EEPACC_OUT_N; ++f) if (f == lane) val = so.out[f];
        dst[(size_t)lane * stride_field] = val;
    }
}

// B2: one step for B instances
template <int MMAX, int NS, int WPB>
__global__ void __launch_bounds__(64 * WPB)
k_fbs_step(fbs_step_args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevCfg& C = *a.cfg;
    rc_table_init<MMAX>();
    const int b = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (b >= a.B) return;
    double* Hs;
    FMem<MMAX, NS>& M = *wave_mem<MMAX, NS>(smem, Hs);
    double* Hb = a.hb + ((size_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * (NS * NS);
    const int lane = lane_id(), N = C.N;
    double* stt = a.state + (size_t)b * kFbsStateDoubles;
    StepIn in{a.s[b], a.v[b], a.a_prev[b], a.t0[b], a.s_tv[b], a.v_tv[b], a.a_tv_prev[b], 0.0, a.k_step, 0};
    unsigned long long code = reinterpret_cast<unsigned long long*>(stt)[lane];
    double A22 = st_A22(stt)[lane], D2 = st_D2(stt)[lane];
    const int idx = lane < N ? lane + 1 : N;
    const double ps = st_sp(stt)[idx], pv = st_vp(stt)[idx];
    StepOut so;
    double sp, vp;
    fb_step<MMAX, NS>(C, M, Hs, Hb, in, A22, D2, code, so, sp, vp, ps, pv);
    st_A22(stt)[lane] = A22; st_D2(stt)[lane] = D2;
    if (lane <= N) { st_sp(stt)[lane] = sp; st_vp(stt)[lane] = vp; }
    reinterpret_cast<unsigned long long*>(stt)[lane] = shift_codes(code, N);
    write_out(a.out + b, (size_t)a.B, so, lane);
    if (a.s_pred && lane <= N) a.s_pred[(size_t)lane * a.B + b] = sp;
    if (a.v_pred && lane <= N) a.v_pred[(size_t)lane * a.B + b] = vp;
    if (lane == 0) { a.status[b] = so.status; if (a.iters) a.iters[b] = so.iters; }
}

// B1: closed loop over n_steps for B instances (ABO/RunOpt_FBMPC.m:161-331); work units (instance, chunk of MPC
// steps) handed out through a device-wide counter as in the ABMPC kernel (eepacc_ab_impl.inc, k_run_abmpc)
template <int MMAX, int NS, int WPB>
__global__ void __launch_bounds__(64 * WPB, ((NS <= 32 && WPB <= 2) ? 2 : 1))
k_fbs_run(fbs_run_args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevCfg& C = *a.cfg;
    rc_table_init<MMAX>();
    double* Hs;
    FMem<MMAX, NS>& M = *wave_mem<MMAX, NS>(smem, Hs);
    double* Hb = a.hb + ((size_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * (NS * NS);
    const int lane = lane_id(), N = C.N, B = a.B;
    const double Ts = C.Tvec[0];
    const int n_chunks = (a.n_steps + a.chunk_steps - 1) / a.chunk_steps;
    const int n_units = n_chunks * B;
    for (int fetch = 0; fetch <= n_units; ++fetch) {
        int u = 0;
        if (lane == 0) u = atomicAdd(a.work_counter, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= n_units || u < 0) break;
        const int chunk = u / B, b = u - chunk * B;
        const int kk0 = chunk * a.chunk_steps;
        const int kk1 = (kk0 + a.chunk_steps < a.n_steps) ? kk0 + a.chunk_steps : a.n_steps;
        bool failed = false;
        if (chunk > 0) {
            int spins = 0;
            while (__hip_atomic_load(&a.done[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunk) {
                __builtin_amdgcn_s_sleep(32);
                if (++spins > a.spin_limit) { failed = true; break; }
                if ((spins & 63) == 0 && __hip_atomic_load(a.err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (failed || __hip_atomic_load(a.err_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            if (lane == 0) {
                if (failed) atomicOr(a.err_word, 1);
                for (int kk = kk0; kk < kk1; ++kk) a.status[(size_t)kk * B + b] = 3;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_max(&a.done[b], chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        double* stt = a.state + (size_t)b * kFbsStateDoubles;
        unsigned long long code = 0ull;
        double A22 = 1.0, D2 = 0.0;
        double s_prev = 0, v_prev = 0, Fm_prev = 0, Fb_prev = 0, v_tv_measured = 0.0, t_0 = 0.0;
        const int idx = lane < N ? lane + 1 : N;
        double ps = 0.0, pv = 0.0;
        if (a.k_start + kk0 > 0) {
            s_prev = a.carry[0 * (size_t)B + b]; v_prev = a.carry[1 * (size_t)B + b];
            Fm_prev = a.carry[2 * (size_t)B + b]; Fb_prev = a.carry[3 * (size_t)B + b];
            v_tv_measured = a.carry[4 * (size_t)B + b]; t_0 = a.carry[5 * (size_t)B + b];
            code = reinterpret_cast<unsigned long long*>(stt)[lane];
            A22 = st_A22(stt)[lane]; D2 = st_D2(stt)[lane];
            ps = st_sp(stt)[idx]; pv = st_vp(stt)[idx];
        }
        int it_total = 0;
        for (int kk = kk0; kk < kk1; ++kk) {
            StepIn in;
            in.k_step = a.k_start + kk;
            if (in.k_step == 0) {                                // :165-183
                in.s = a.s0[b]; in.v = a.v0[b]; in.a_prev = a.a_m1[b];
                in.s_tv = a.s_tv[b]; in.v_tv = 0.0; in.a_tv_prev = 0.0;
                v_tv_measured = 0.0;
                in.v_prev = 0.0; in.have_vprev = 0;
            } else {                                             // :184-200
                double sm, vm;
                plant_rk4(C, s_prev, v_prev, Fm_prev + Fb_prev, sm, vm);
                in.s = sm; in.v = vm;
                in.a_prev = (vm - v_prev) / Ts;
                in.s_tv = a.s_tv[(size_t)kk * B + b];
                const double v_tv_prev = v_tv_measured;
                v_tv_measured = a.v_tv[(size_t)kk * B + b];
                in.v_tv = v_tv_measured;
                in.a_tv_prev = (v_tv_measured - v_tv_prev) / Ts;
                in.v_prev = v_prev; in.have_vprev = 1;
            }
            in.t0 = t_0;
            StepOut so;
            double sp, vp;
            fb_step<MMAX, NS>(C, M, Hs, Hb, in, A22, D2, code, so, sp, vp, ps, pv);
            if (C.paramEstSetting == 2) {
                ps = __shfl(sp, idx, 64); pv = __shfl(vp, idx, 64);
            }
            if (kk == kk1 - 1 && lane <= N) { st_sp(stt)[lane] = sp; st_vp(stt)[lane] = vp; }
            code = a.cold ? 0ull : shift_codes(code, N);
            write_out(a.traj + (size_t)kk * EEPACC_OUT_N * B + b, (size_t)B, so, lane);
            if (lane == 0) a.status[(size_t)kk * B + b] = so.status;
            it_total += so.iters;
            s_prev = so.out[EEPACC_OUT_S]; v_prev = so.out[EEPACC_OUT_V];
            Fm_prev = so.out[EEPACC_OUT_FM]; Fb_prev = so.out[EEPACC_OUT_FB];
            t_0 += Ts;                                           // :321
        }
        reinterpret_cast<unsigned long long*>(stt)[lane] = code;
        st_A22(stt)[lane] = A22; st_D2(stt)[lane] = D2;
        if (lane == 0) {
            a.carry[0 * (size_t)B + b] = s_prev; a.carry[1 * (size_t)B + b] = v_prev;
            a.carry[2 * (size_t)B + b] = Fm_prev; a.carry[3 * (size_t)B + b] = Fb_prev;
            a.carry[4 * (size_t)B + b] = v_tv_measured; a.carry[5 * (size_t)B + b] = t_0;
            if (a.iters_total) atomicAdd(&a.iters_total[b], it_total);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_max(&a.done[b], chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace fbs

#ifdef EEPACC_FBS_TIMING
extern "C" int eepacc_debug_fbs_prof(unsigned long long* out, int reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fbs::g_fbs_prof), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(fbs::g_fbs_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

// ----------------------------------------------------------------------------------------------
// host-side launchers used by eepacc_capi.cpp
constexpr int kFMMaxSmall = 32, kFNSSmall = 32, kFWpbSmall = 7;     // N <= 32: 21.7 KB of LDS per wave, one block of 7 waves per CU
                                                                    // (working-set rows are independent in the N-dimensional u-space: m <= N)
constexpr int kFMMaxLarge = 66, kFNSLarge = 64, kFWpbLarge = 1;     // N <= 63

// settings the structured solver represents; everything else goes through the dense path (eepacc_fb.hip)
bool fbs_supported(const DevCfg& C) {
    if (C.mb_any) return false;                 // blocked moves (CreateQP_FB.m:346-356) keep their equality rows
    if (C.b_quadr[3] != 0.0) return false;      // Fm^2 term of the power fit would give w its own curvature
    const double Kr = (30.0 / 3.14159265358979323846) * C.phi;
    if (!(C.fb_w[0] * Kr * C.b_quadr[4] > 0.0)) return false;   // price of the friction-brake share must rise with speed
    if (!(C.fb_w[4] > 0.0) || C.fb_w[3] < 0.0 || C.fb_w[5] < 0.0 || C.fb_w[6] < 0.0 || !(C.fb_w[1] > 0.0)) return false;
    return true;
}

size_t fbs_smem_bytes(int N) {
    return N <= kFNSSmall ? fbs::wave_bytes(sizeof(fbs::FMem<kFMMaxSmall, kFNSSmall>), kFNSSmall) * kFWpbSmall
                          : fbs::wave_bytes(sizeof(fbs::FMem<kFMMaxLarge, kFNSLarge>), kFNSLarge) * kFWpbLarge;
}

// dynamic LDS a block may ask for: 160 KB per CU minus the kernels' static index table (one ushort per packed entry of P)
constexpr size_t kFbsLdsBudget = 160 * 1024 - 4608;

static int fbs_run_grid(int N, int n_units, int num_cus) {
    const size_t smem = fbs_smem_bytes(N);
    int per_cu = (int)((kFbsLdsBudget) / smem);
    if (per_cu < 1) per_cu = 1;
    const int wpb = N > kFNSSmall ? kFWpbLarge : kFWpbSmall;
    int grid = num_cus * per_cu;
    const int need = (n_units + wpb - 1) / wpb;
    return grid > need ? need : grid;
}

// scratch for the base inverse of every wave a launch over B instances can have (step: one wave per instance;
// closed loop: at most the chip-filling grid)
size_t fbs_hb_doubles(int N, int B, int num_cus) {
    const int ns = N > kFNSSmall ? kFNSLarge : kFNSSmall, wpb = N > kFNSSmall ? kFWpbLarge : kFWpbSmall;
    const size_t step_waves = (size_t)((B + wpb - 1) / wpb) * wpb;
    const size_t run_waves = (size_t)fbs_run_grid(N, 0x7fffffff / 2, num_cus) * wpb;
    return (step_waves > run_waves ? step_waves : run_waves) * ns * ns;
}

hipError_t fbs_set_max_smem() {
    const void* fns[4] = {reinterpret_cast<const void*>(&fbs::k_fbs_step<kFMMaxSmall, kFNSSmall, kFWpbSmall>),
                          reinterpret_cast<const void*>(&fbs::k_fbs_step<kFMMaxLarge, kFNSLarge, kFWpbLarge>),
                          reinterpret_cast<const void*>(&fbs::k_fbs_run<kFMMaxSmall, kFNSSmall, kFWpbSmall>),
                          reinterpret_cast<const void*>(&fbs::k_fbs_run<kFMMaxLarge, kFNSLarge, kFWpbLarge>)};
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFbsLdsBudget);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_fbs_step(const fbs_step_args& a, int N, hipStream_t stream) {
    const size_t smem = fbs_smem_bytes(N);
    if (N > kFNSSmall)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(fbs::k_fbs_step<kFMMaxLarge, kFNSLarge, kFWpbLarge>), dim3((a.B + kFWpbLarge - 1) / kFWpbLarge),
                           dim3(64 * kFWpbLarge), smem, stream, a);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(fbs::k_fbs_step<kFMMaxSmall, kFNSSmall, kFWpbSmall>), dim3((a.B + kFWpbSmall - 1) / kFWpbSmall),
                           dim3(64 * kFWpbSmall), smem, stream, a);
    return hipGetLastError();
}

hipError_t launch_fbs_run(const fbs_run_args& a, int N, int num_cus, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(a.work_counter, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.done, 0, sizeof(int) * (size_t)a.B, stream);
    if (e != hipSuccess) return e;
    if (a.iters_total) {
        e = hipMemsetAsync(a.iters_total, 0, sizeof(int32_t) * (size_t)a.B, stream);
        if (e != hipSuccess) return e;
    }
    const size_t smem = fbs_smem_bytes(N);
    const int n_units = ((a.n_steps + a.chunk_steps - 1) / a.chunk_steps) * a.B;
    const int grid = fbs_run_grid(N, n_units, num_cus);
    if (N > kFNSSmall)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(fbs::k_fbs_run<kFMMaxLarge, kFNSLarge, kFWpbLarge>), dim3(grid), dim3(64 * kFWpbLarge), smem, stream, a);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(fbs::k_fbs_run<kFMMaxSmall, kFNSSmall, kFWpbSmall>), dim3(grid), dim3(64 * kFWpbSmall), smem, stream, a);
    return hipGetLastError();
}

}  // namespace eepacc
