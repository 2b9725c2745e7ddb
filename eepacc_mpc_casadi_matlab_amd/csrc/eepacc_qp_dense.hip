// eepacc_qp_dense.hip -- batched dense QP operator (boundary level B3):
//     sol = QPsolver('h',H,'g',c,'a',G,'lbx',..,'ubx',..,'lba',g_lb,'uba',g_ub)
// (ABO/RunOpt_ABMPC.m:252, ABO/RunOpt_FBMPC.m:278; CasADi conic, CAS/+casadi/conic.m:951-966)
//     min 1/2 x'Hx + g'x   s.t.  lba <= A x <= uba,  lbx <= x <= ubx     (+-inf = absent)
//
// One QP per 256-thread workgroup (4 wavefronts); workgroups are persistent and walk the batch.
// Matrices live in a per-workgroup global workspace (they do not fit LDS for nV >= 150) that
// stays L2/MALL resident; every vector of the iteration lives in LDS.
//
// Method (handles the PSD Hessian of the AB QP and the indefinite one of the FB QP):
//   proximal outer loop on H + rho I (rho raised until the Cholesky factor exists),
//   Goldfarb-Idnani dual active set inside, with J = L^-T Q kept TRANSPOSED (JT[j][k]=J[k][j],
//   so every sweep over a J column is a coalesced row read), R column-major, constraint
//   addition by one Householder reflector applied by all threads (no serial Givens chain),
//   removal by Givens on two contiguous JT rows; closed-form crash start on the bounds of
//   curvature-free variables; exact KKT solve (LU, partial pivoting, iterative refinement) on
//   the final working set with the unregularised H, and KKT verification.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "eepacc_qp_dense.h"

#ifdef EEPACC_QP_TIMING
__device__ long long g_qp_prof[16];
#define TIC(var) long long var = wall_clock64()
#define TOC(var, slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_qp_prof[slot] += wall_clock64() - var; } while (0)
#else
#define TIC(var)
#define TOC(var, slot)
#endif

namespace {

constexpr int QT = 256;
constexpr int QW = QT / 64;

struct Prob {            // one instance
    int n, nC, m1;
    const double *H, *g, *A, *lba, *uba, *lbx, *ubx;
};

struct Lds {
    double *x, *xc, *gr, *np, *d, *z, *r, *u, *xp, *up, *t, *hv, *rhs, *sol, *res, *fcol, *ax, *red;
    int *act, *crash, *piv, *ired;
    int* jend;              // per row of A: one past its last non-zero column
    int* svar;              // per row of A: its column if it has exactly one non-zero, else -1
    int *kk_kind, *kk_vmap, *kk_free, *kk_gpos;   // index maps of kkt_solve
    int* live;              // columns of A the violation scan has to read (ascending)
    unsigned char* is_act;
};

struct Ws {              // per-workgroup global workspace
    double *Hs, *J0T, *JT, *R, *R2, *T, *T2, *K;   // n^2 each, K (2n+2)^2 ; L aliases K.  T = R^-1 (column-major)
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ double block_sum(double v, const Lds& S) {
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) S.red[wave_id()] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < QW; ++w) s += S.red[w];
    return s;
}

__device__ double block_max(double v, const Lds& S) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane_id() == 0) S.red[wave_id()] = v;
    __syncthreads();
    double s = S.red[0];
#pragma unroll
    for (int w = 1; w < QW; ++w) s = fmax(s, S.red[w]);
    return s;
}

// arg-min with smallest index on ties; idx < 0 means "none"
__device__ void block_argmin(double& v, int& idx, const Lds& S) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o);
        int oi = __shfl_xor(idx, o);
        bool take = (oi >= 0) && (idx < 0 || ov < v || (ov == v && oi < idx));
        if (take) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (lane_id() == 0) { S.red[wave_id()] = v; S.ired[wave_id()] = idx; }
    __syncthreads();
    v = S.red[0]; idx = S.ired[0];
#pragma unroll
    for (int w = 1; w < QW; ++w) {
        double ov = S.red[w]; int oi = S.ired[w];
        bool take = (oi >= 0) && (idx < 0 || ov < v || (ov == v && oi < idx));
        if (take) { v = ov; idx = oi; }
    }
}

// one-sided constraint c of the list  sgn * (row or variable) >= b
__device__ __forceinline__ bool os_get(const Prob& P, int c, int& row, double& sgn, double& b) {
    if (c < 2 * P.nC) {
        row = c >> 1;
        if ((c & 1) == 0) { if (!P.lba) return false; b = P.lba[row]; sgn = 1.0; }
        else              { if (!P.uba) return false; b = -P.uba[row]; sgn = -1.0; }
    } else {
        int cc = c - 2 * P.nC;
        row = -(cc >> 1) - 1;
        if ((cc & 1) == 0) { if (!P.lbx) return false; b = P.lbx[cc >> 1]; sgn = 1.0; }
        else               { if (!P.ubx) return false; b = -P.ubx[cc >> 1]; sgn = -1.0; }
    }
    return isfinite(b);
}

// np <- normal of one-sided constraint c  (all threads)
__device__ void get_normal(const Prob& P, int c, double* np) {
    int row; double sgn, b;
    os_get(P, c, row, sgn, b);
    for (int j = threadIdx.x; j < P.n; j += QT)
        np[j] = row >= 0 ? sgn * P.A[(size_t)j * P.nC + row] : (j == -row - 1 ? sgn : 0.0);
    __syncthreads();
}

// value of constraint c at x (uniform result)
__device__ double con_value(const Prob& P, int c, const double* x, const Lds& S) {
    int row; double sgn, b;
    os_get(P, c, row, sgn, b);
    if (row < 0) return sgn * x[-row - 1] - b;
    double s = 0.0;
    for (int j = threadIdx.x; j < P.n; j += QT) s += P.A[(size_t)j * P.nC + row] * x[j];
    s = block_sum(s, S);
    return sgn * s - b;
}

// ax = A x  (thread per row; loads issued eight at a time, summation order j = 0..).  jend[i] is one
// past the last non-zero of row i (condensed MPC rows only reach back over earlier stages): the
// columns beyond it hold exact zeros and are not read.
__device__ void rows_times(const Prob& P, const double* x, double* ax, const int* jend) {
    const size_t ld = (size_t)P.nC;
    for (int i = threadIdx.x; i < P.nC; i += QT) {
        double s = 0.0;
        const double* a = P.A + i;
        const int je = jend[i];
        int j = 0;
        for (; j + 8 <= je; j += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a[(size_t)(j + u) * ld];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u] * x[j + u];
        }
        for (; j < je; ++j) s += a[(size_t)j * ld] * x[j];
        ax[i] = s;
    }
    __syncthreads();
}

// the same product restricted to the columns live[0..nlive): variables held at zero by an active bound
// (most slack variables of an MPC QP) contribute nothing and their columns are not read
__device__ void rows_times_live(const Prob& P, const double* x, double* ax, const int* jend, const int* live, int nlive) {
    const size_t ld = (size_t)P.nC;
    for (int i = threadIdx.x; i < P.nC; i += QT) {
        double s = 0.0;
        const double* a = P.A + i;
        const int je = jend[i];
        int t = 0;
        for (; t + 8 <= nlive && live[t + 7] < je; t += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a[(size_t)live[t + u] * ld];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u] * x[live[t + u]];
        }
        for (; t < nlive && live[t] < je; ++t) s += a[(size_t)live[t] * ld] * x[live[t]];
        ax[i] = s;
    }
    __syncthreads();
}

// dv[j] = sum_k M[j][k] v[k]  (row-major n x n in global, one wavefront per row, four rows in flight)
__device__ void rowdot(const double* M, int n, const double* v, double* out) {
    for (int j0 = 4 * wave_id(); j0 < n; j0 += 4 * QW) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = lane_id(); k < n; k += 64) {
            double m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) m[u] = (j0 + u < n) ? M[(size_t)(j0 + u) * n + k] : 0.0;
            const double vk = v[k];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += m[u] * vk;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double r = wave_sum(s[u]);
            if (lane_id() == 0 && j0 + u < n) out[j0 + u] = r;
        }
    }
    __syncthreads();
}

// out[k] = sum_{j=j0}^{n-1} M[j][k] c[j]  (thread per k)
__device__ void coldot(const double* M, int n, int j0, const double* c, double* out, double scale) {
    for (int k = threadIdx.x; k < n; k += QT) {
        double s = 0.0;
        int j = j0;
        for (; j + 8 <= n; j += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = M[(size_t)(j + u) * n + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u] * c[j + u];
        }
        for (; j < n; ++j) s += M[(size_t)j * n + k] * c[j];
        out[k] = scale * s;
    }
    __syncthreads();
}

// Cholesky (lower, in place, row-major).  Uniform return: 0 ok, -1 not positive definite.
__device__ int chol_lower(double* L, int n, const Lds& S) {
    for (int j = 0; j < n; ++j) {
        double djj = L[(size_t)j * n + j];
        if (!(djj > 0.0)) return -1;
        double l = sqrt(djj);
        __syncthreads();
        for (int i = j + threadIdx.x; i < n; i += QT) {
            double v = (i == j) ? l : L[(size_t)i * n + j] / l;
            L[(size_t)i * n + j] = v;
            S.fcol[i] = v;
        }
        __syncthreads();
        for (int i = j + 1 + wave_id(); i < n; i += QW) {
            const double fi = S.fcol[i];
            for (int k = j + 1 + lane_id(); k <= i; k += 64) L[(size_t)i * n + k] -= fi * S.fcol[k];
        }
        __syncthreads();
    }
    return 0;
}

// X = L^-1 (lower triangular, row-major) by column-oriented forward substitution on the rows of X:
// row k is final after its division; it is then eliminated from all later rows (a rank-one update
// of the (n-k-1) x (k+1) block, rows over the wavefronts, columns over the lanes).
__device__ void tri_inverse(const double* L, double* X, int n, const Lds& S) {
    for (int i = wave_id(); i < n; i += QW)
        for (int c = lane_id(); c < n; c += 64) X[(size_t)i * n + c] = (i == c) ? 1.0 : 0.0;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const double dk = L[(size_t)k * n + k];
        for (int c = threadIdx.x; c <= k; c += QT) {
            const double v = X[(size_t)k * n + c] / dk;
            X[(size_t)k * n + c] = v;
            S.fcol[c] = v;
        }
        __syncthreads();
        for (int i = k + 1 + wave_id(); i < n; i += QW) {
            const double lik = L[(size_t)i * n + k];
            double* xi = X + (size_t)i * n;
            for (int c = lane_id(); c <= k; c += 64) xi[c] -= lik * S.fcol[c];
        }
        __syncthreads();
    }
}

// ---- LU with partial pivoting on K (Nk x Nk row-major, global).  returns min |pivot|
__device__ double lu_factor(double* K, int Nk, const Lds& S) {
    double minpiv = INFINITY;
    for (int k = 0; k < Nk; ++k) {
        double best = -1.0; int p = -1;
        for (int i = k + threadIdx.x; i < Nk; i += QT) {
            double v = fabs(K[(size_t)i * Nk + k]);
            if (v > best) { best = v; p = i; }
        }
        double nb = -best;                 // arg-max as arg-min of the negative, first index on ties
        block_argmin(nb, p, S);
        best = -nb;
        if (threadIdx.x == 0) S.piv[k] = p;
        if (best < minpiv) minpiv = best;
        if (best == 0.0) return 0.0;
        if (p != k)
            for (int j = threadIdx.x; j < Nk; j += QT) {
                double t = K[(size_t)k * Nk + j];
                K[(size_t)k * Nk + j] = K[(size_t)p * Nk + j];
                K[(size_t)p * Nk + j] = t;
            }
        __syncthreads();
        double inv = 1.0 / K[(size_t)k * Nk + k];
        for (int i = k + 1 + threadIdx.x; i < Nk; i += QT) {
            double f = K[(size_t)i * Nk + k] * inv;
            K[(size_t)i * Nk + k] = f;
            S.fcol[i] = f;
        }
        __syncthreads();
        {
            // pivot row in registers (up to 12 entries per lane), rows of the trailing block over the waves
            double pr[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) { int j = k + 1 + lane_id() + 64 * u; pr[u] = j < Nk ? K[(size_t)k * Nk + j] : 0.0; }
            for (int i = k + 1 + wave_id(); i < Nk; i += QW) {
                const double f = S.fcol[i];
                if (f == 0.0) continue;
                double* ri = K + (size_t)i * Nk;
#pragma unroll
                for (int u = 0; u < 12; ++u) { int j = k + 1 + lane_id() + 64 * u; if (j < Nk) ri[j] -= f * pr[u]; }
            }
        }
        __syncthreads();
    }
    return minpiv;
}

__device__ void lu_solve(const double* LU, int Nk, double* b, const Lds& S) {
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < Nk; ++k) {
            int p = S.piv[k];
            if (p != k) { double t = b[k]; b[k] = b[p]; b[p] = t; }
        }
    __syncthreads();
    for (int k = 0; k < Nk; ++k) {
        double bk = b[k];
        __syncthreads();
        for (int i = k + 1 + threadIdx.x; i < Nk; i += QT) b[i] -= LU[(size_t)i * Nk + k] * bk;
        __syncthreads();
    }
    for (int i = Nk - 1; i >= 0; --i) {
        double bi = b[i] / LU[(size_t)i * Nk + i];
        __syncthreads();
        if (threadIdx.x == 0) b[i] = bi;
        for (int r = threadIdx.x; r < i; r += QT) b[r] -= LU[(size_t)r * Nk + i] * bi;
        __syncthreads();
    }
}

// residual of the full KKT system at S.sol = [x; y] (y = -u, one entry per working-set row):
//   res[i]   = -gv[i] - (Hs + rho I) x - sum_c N[c][i] y_c      (i < n; one thread per variable)
//   res[n+c] = b_c - n_c' x                                      (from ax = A x, left in S.ax)
__device__ void kkt_residual(const Prob& P, const Ws& W, const Lds& S, double rho, const double* gv,
                             const int* act, int q) {
    const int n = P.n;
    rows_times(P, S.sol, S.ax, S.jend);
    for (int i = threadIdx.x; i < n; i += QT) {
        double s = -gv[i] - rho * S.sol[i];
        const double* h = W.Hs + (size_t)i * n;
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = h[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fma(-v[u], S.sol[j + u], s);
        }
        for (; j < n; ++j) s = fma(-h[j], S.sol[j], s);
        const double* acol = P.A + (size_t)i * P.nC;
        for (int c = 0; c < q; ++c) {
            const int id = act[c];
            const double sgn = (id & 1) ? -1.0 : 1.0;
            if (id >= 2 * P.nC) {
                if (((id - 2 * P.nC) >> 1) == i) s = fma(-sgn, S.sol[n + c], s);
            } else {
                const int row = id >> 1, sv = S.svar[row];
                if (sv >= 0 ? sv == i : i < S.jend[row]) s = fma(-sgn * acol[row], S.sol[n + c], s);
            }
        }
        S.res[i] = s;
    }
    for (int c = threadIdx.x; c < q; c += QT) {
        int row; double sgn, b;
        os_get(P, act[c], row, sgn, b);
        S.res[n + c] = b - sgn * (row >= 0 ? S.ax[row] : S.sol[-row - 1]);
    }
    __syncthreads();
}

// Exact KKT solve on the working set act[0..q): [Hm N'; N 0][x; -u] = [-gv; b], Hm = Hs + rho I.
// Working-set rows with a single non-zero (slack and variable bounds -- two thirds of a typical
// MPC working set) fix their variable directly; LU with partial pivoting runs on the remaining
// (free variables + general rows) system only, followed by two rounds of iterative refinement
// against residuals evaluated from the problem data.  Outputs xo[n], uo[q]; kkt[3] (uniform).
// returns 0 / -1 (singular).
__device__ int kkt_solve(const Prob& P, const Ws& W, const Lds& S, double rho, const double* gv,
                         const int* act, int q, double* xo, double* uo, double kkt[3]) {
    const int n = P.n;
    double* K = W.K;
    int* kind = S.kk_kind;      // per working-set row: variable it fixes, or -1
    int* vmap = S.kk_vmap;      // per variable: index among the free ones, or -1
    int* freev = S.kk_free;     // free variables
    int* gpos = S.kk_gpos;      // general working-set rows (positions in act)
    __syncthreads();
    for (int i = threadIdx.x; i < n + q; i += QT) S.sol[i] = 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int v = 0; v < n; ++v) vmap[v] = 0;
        int qg = 0;
        for (int c = 0; c < q; ++c) {
            int row; double sgn, b;
            os_get(P, act[c], row, sgn, b);
            const int v = row < 0 ? -row - 1 : S.svar[row];
            if (v >= 0 && vmap[v] == 0) {
                const double a = row < 0 ? sgn : sgn * P.A[(size_t)v * P.nC + row];
                kind[c] = v; vmap[v] = -1;
                S.sol[v] = b / a;
            } else { kind[c] = -1; gpos[qg++] = c; }
        }
        int nr = 0;
        for (int v = 0; v < n; ++v) if (vmap[v] == 0) { vmap[v] = nr; freev[nr++] = v; }
        S.ired[2] = nr; S.ired[3] = qg;
    }
    __syncthreads();
    const int nr = S.ired[2], qg = S.ired[3], Nr = nr + qg;
    // right-hand side of the reduced system = full residual at (x_fixed, 0, y = 0)
    kkt_residual(P, W, S, rho, gv, act, q);
    // reduced matrix
    for (int i = wave_id(); i < Nr; i += QW) {
        double* kr = K + (size_t)i * Nr;
        if (i < nr) {
            const double* h = W.Hs + (size_t)freev[i] * n;
            for (int j = lane_id(); j < nr; j += 64) kr[j] = h[freev[j]] + (i == j ? rho : 0.0);
        } else {
            int row; double sgn, b;
            os_get(P, act[gpos[i - nr]], row, sgn, b);
            for (int j = lane_id(); j < nr; j += 64) {
                const int v = freev[j];
                const double a = row >= 0 ? sgn * P.A[(size_t)v * P.nC + row] : (v == -row - 1 ? sgn : 0.0);
                kr[j] = a;
                K[(size_t)j * Nr + i] = a;
            }
            for (int j = nr + lane_id(); j < Nr; j += 64) kr[j] = 0.0;
        }
    }
    __syncthreads();
    double minpiv = Nr > 0 ? lu_factor(K, Nr, S) : 1.0;
    if (minpiv < 1e-13) return -1;
    for (int round = 0; round < 3; ++round) {
        // gather the reduced residual, solve, scatter the correction
        for (int i = threadIdx.x; i < Nr; i += QT) S.rhs[i] = i < nr ? S.res[freev[i]] : S.res[n + gpos[i - nr]];
        if (Nr > 0) lu_solve(K, Nr, S.rhs, S);
        __syncthreads();
        for (int i = threadIdx.x; i < Nr; i += QT) {
            if (i < nr) S.sol[freev[i]] += S.rhs[i]; else S.sol[n + gpos[i - nr]] += S.rhs[i];
        }
        // multipliers of the fixing rows from the stationarity rows of their variables
        for (int c = threadIdx.x; c < q; c += QT) if (kind[c] >= 0) S.sol[n + c] = 0.0;
        __syncthreads();
        kkt_residual(P, W, S, rho, gv, act, q);
        for (int c = threadIdx.x; c < q; c += QT)
            if (kind[c] >= 0) {
                int row; double sgn, b;
                os_get(P, act[c], row, sgn, b);
                const int v = kind[c];
                const double a = row < 0 ? sgn : sgn * P.A[(size_t)v * P.nC + row];
                S.sol[n + c] = S.res[v] / a;
            }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += QT) xo[i] = S.sol[i];
    for (int c = threadIdx.x; c < q; c += QT) uo[c] = -S.sol[n + c];
    __syncthreads();
    // verification: stationarity (relative), worst primal violation over all rows, most negative
    // multiplier.  The last kkt_residual call was made at this x with the fixing rows' multipliers
    // zeroed: its rows of free variables are the stationarity residual (the rows of fixed variables
    // vanish by the construction of those multipliers), and S.ax = A x.
    double stat = 0.0, scale = 1.0, dneg = 0.0, pviol = 0.0;
    for (int i = threadIdx.x; i < n; i += QT) {
        if (vmap[i] >= 0) stat = fmax(stat, fabs(S.res[i]));
        scale = fmax(scale, fabs(gv[i]));
    }
    for (int c = threadIdx.x; c < q; c += QT) { dneg = fmax(dneg, -uo[c]); scale = fmax(scale, fabs(uo[c])); }
    for (int c = threadIdx.x; c < P.m1; c += QT) {
        int row; double sgn, b;
        if (!os_get(P, c, row, sgn, b)) continue;
        double v = (sgn * (row >= 0 ? S.ax[row] : xo[-row - 1]) - b) / (1.0 + fabs(b));
        pviol = fmax(pviol, -v);
    }
    stat = block_max(stat, S); scale = block_max(scale, S); dneg = block_max(dneg, S); pviol = block_max(pviol, S);
    kkt[0] = stat / scale; kkt[1] = pviol; kkt[2] = dneg / scale;
    return 0;
}

// ---- Goldfarb-Idnani working-set updates on (JT, R) -------------------------------------
// add: S.d = J' np already computed.  Returns 0 ok / -1 dependent.  q is NOT incremented here.
__device__ int gi_add(const Ws& W, const Lds& S, int n, int q) {
    double* d = S.d;
    double part = 0.0;
    for (int j = q + threadIdx.x; j < n; j += QT) part += d[j] * d[j];
    double nrm2 = block_sum(part, S);
    double nrm = sqrt(nrm2);
    double d0 = d[0], dq = d[q];
    if (nrm <= 1e-14 * (1.0 + fabs(d0))) {
        for (int i = threadIdx.x; i <= q; i += QT) W.R[(size_t)q * n + i] = (i < q) ? d[i] : nrm;
        __syncthreads();
        return -1;
    }
    double alpha = dq >= 0.0 ? -nrm : nrm;
    double v0 = dq - alpha;
    // v'v = nrm2 - dq^2 + v0^2
    double vtv = nrm2 - dq * dq + v0 * v0;
    double beta = 2.0 / vtv;
    __syncthreads();
    if (threadIdx.x == 0) d[q] = v0;           // d[q..n) now holds the reflector v
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += QT) {
        double t = 0.0;
        for (int j = q; j < n; ++j) t += W.JT[(size_t)j * n + k] * d[j];
        t *= beta;
        for (int j = q; j < n; ++j) W.JT[(size_t)j * n + k] -= t * d[j];
    }
    for (int i = threadIdx.x; i <= q; i += QT) {
        W.R[(size_t)q * n + i] = (i < q) ? d[i] : alpha;
        // inverse of [R d1; 0 alpha] : last column (-R^-1 d1 / alpha ; 1/alpha), and S.r = R^-1 d1
        W.T[(size_t)q * n + i] = (i < q) ? -S.r[i] / alpha : 1.0 / alpha;
    }
    __syncthreads();
    return 0;
}

// one chain of plane rotations along a strided vector: v[j], v[j+1] <- rot_j, j = l .. qn-1
// (v[j] = base[j*ld]).  The right-hand elements are loaded four steps ahead of their use.
__device__ __forceinline__ void rot_chain(double* base, size_t ld, int l, int qn, const double* Cs, const double* Sn,
                                          bool keep_last) {
    double x = base[(size_t)l * ld];
    int j = l;
    for (; j + 4 <= qn; j += 4) {
        double y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = base[(size_t)(j + 1 + u) * ld];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double c = Cs[j + u], sn = Sn[j + u];
            base[(size_t)(j + u) * ld] = c * x + sn * y[u];
            x = -sn * x + c * y[u];
        }
    }
    for (; j < qn; ++j) {
        const double y = base[(size_t)(j + 1) * ld];
        const double c = Cs[j], sn = Sn[j];
        base[(size_t)j * ld] = c * x + sn * y;
        x = -sn * x + c * y;
    }
    if (keep_last) base[(size_t)qn * ld] = x;
}

// drop active constraint at position l (q = count before the drop).
// Wavefront 0 re-triangularises R (column l removed) with the row pair it works on held in
// registers -- no barrier inside the chain -- and records the rotations; meanwhile the other
// wavefronts copy the untouched part of R and strip row l from T = R^-1.  Then every thread applies
// the recorded chain to one column of J (a row pair of JT per step) or one row of T.
__device__ void gi_drop(Ws& W, const Lds& S, int n, int q, int l) {
    __syncthreads();
    const int qn = q - 1;
    double* Cs = S.res;
    double* Sn = S.fcol;
    if (wave_id() == 0) {
        const int lane = lane_id();
        double carry[6], y[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int kk = l + lane + 64 * u;
            carry[u] = kk < qn ? W.R[(size_t)(kk + 1) * n + l] : 0.0;
            y[u] = (kk < qn && l < qn) ? W.R[(size_t)(kk + 1) * n + l + 1] : 0.0;
        }
        for (int j = l; j < qn; ++j) {
            double yn[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int kk = l + lane + 64 * u;
                yn[u] = (kk < qn && kk >= j + 1 && j + 1 < qn) ? W.R[(size_t)(kk + 1) * n + j + 2] : 0.0;
            }
            const int oj = j - l, ol = oj & 63, ou = oj >> 6;
            double av = ou == 0 ? carry[0] : ou == 1 ? carry[1] : ou == 2 ? carry[2] : ou == 3 ? carry[3] : ou == 4 ? carry[4] : carry[5];
            double bv = ou == 0 ? y[0] : ou == 1 ? y[1] : ou == 2 ? y[2] : ou == 3 ? y[3] : ou == 4 ? y[4] : y[5];
            av = __shfl(av, ol); bv = __shfl(bv, ol);
            double c = 1.0, sn = 0.0;
            if (bv != 0.0) { const double h = hypot(av, bv); c = av / h; sn = bv / h; }
            if (lane == 0) { Cs[j] = c; Sn[j] = sn; }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int kk = l + lane + 64 * u;
                if (kk < qn && kk >= j) {
                    W.R2[(size_t)kk * n + j] = c * carry[u] + sn * y[u];
                    carry[u] = -sn * carry[u] + c * y[u];
                }
                y[u] = yn[u];
            }
        }
    } else {
        const int t = threadIdx.x - 64, nt = QT - 64;
        if (t == 0) {
            for (int j = l; j < q - 1; ++j) { S.act[j] = S.act[j + 1]; S.u[j] = S.u[j + 1]; }
            S.u[q - 1] = S.u[q];
            S.u[q] = 0.0;
        }
        const int w = wave_id() - 1, nw = QW - 1;
        // R2: columns before l unchanged; rows above l of the shifted columns
        for (int k = w; k < qn; k += nw) {
            const int ko = k < l ? k : k + 1;
            const int rows = k < l ? k + 1 : l;
            for (int i = lane_id(); i < rows; i += 64) W.R2[(size_t)k * n + i] = W.R[(size_t)ko * n + i];
        }
        // T2 <- T without row l (the inverse of the re-triangularised R is that matrix with the same
        // rotations applied to its columns, last column dropped)
        for (int k = w; k < q; k += nw)
            for (int i = lane_id(); i < qn; i += 64) {
                const int io = i < l ? i : i + 1;
                W.T2[(size_t)k * n + i] = io <= k ? W.T[(size_t)k * n + io] : 0.0;
            }
        (void)t; (void)nt;
    }
    { double* tmp = W.T; W.T = W.T2; W.T2 = tmp; }
    { double* tmp = W.R; W.R = W.R2; W.R2 = tmp; }
    __syncthreads();
    if (l < qn) {
        for (int task = threadIdx.x; task < n + qn; task += QT) {
            if (task < n) rot_chain(W.JT + task, (size_t)n, l, qn, Cs, Sn, true);
            else rot_chain(W.T + (task - n), (size_t)n, l, qn, Cs, Sn, false);
        }
    }
    __syncthreads();
}

// (x,u) of the working set from the factors at hand: with J'N' = [R;0] and G^-1 = JJ',
//   y = R^-T b,  x = J1 y - J2 (J'gr)_2,  u = R^-1 (y + (J'gr)_1)        -> S.xp, S.up
__device__ void factor_refresh(const Prob& P, const Ws& W, const Lds& S, int n, int q) {
    rowdot(W.JT, n, S.gr, S.t);                                   // t = J' gr
    for (int k = threadIdx.x; k < q; k += QT) {                   // y = T' b
        int row; double sgn, b;
        double acc = 0.0;
        for (int i = 0; i <= k; ++i) {
            os_get(P, S.act[i], row, sgn, b);
            acc += W.T[(size_t)k * n + i] * b;
        }
        S.z[k] = acc;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += QT) S.d[j] = j < q ? S.z[j] : -S.t[j];
    __syncthreads();
    coldot(W.JT, n, 0, S.d, S.xp, 1.0);                           // x
    for (int i = threadIdx.x; i < q; i += QT) {                   // u = T (y + t_1)
        double acc = 0.0;
        for (int k = i; k < q; ++k) acc += W.T[(size_t)k * n + i] * (S.z[k] + S.t[k]);
        S.up[i] = acc;
    }
    __syncthreads();
}

// GI solve of  min 1/2 x'(Hs+rho I)x + gr'x  s.t. list.  Returns 0 ok, 1 infeasible, 2 limit.
// warm: the factors, working set (S.act, S.is_act, q_out) of the previous proximal round are still in
// place; if its multipliers stay non-negative for the new linear term the round starts from there.
__device__ int gi_solve(const Prob& P, Ws& W, const Lds& S, double rho, int n_crash_in,
                        int& q_out, int& iters_out, int max_iter, bool warm) {
    const int n = P.n;
    const int n_crash = n_crash_in;
    int q = 0, iters = 0, status = 0;
    if (warm && q_out > 0) {
        factor_refresh(P, W, S, n, q_out);
        double mn = 0.0;
        for (int j = threadIdx.x; j < q_out; j += QT) mn = fmax(mn, -S.up[j]);
        mn = block_max(mn, S);
        if (mn > 0.0) warm = false;
        else {
            q = q_out;
            for (int i = threadIdx.x; i < n; i += QT) S.x[i] = S.xp[i];
            for (int j = threadIdx.x; j < q; j += QT) S.u[j] = S.up[j];
            __syncthreads();
        }
    } else warm = false;
    if (!warm) {
    for (int c = threadIdx.x; c < P.m1; c += QT) S.is_act[c] = 0;
    __syncthreads();
    // x = -G^-1 gr = -J (J' gr) with J = J0 = L^-T
    rowdot(W.J0T, n, S.gr, S.t);
    coldot(W.J0T, n, 0, S.t, S.x, -1.0);
    TIC(t_cr);
    // Closed-form crash start.  A crash variable has no curvature, so in H + rho I it is decoupled:
    // row and column `var` of L^-1 hold only the diagonal 1/sqrt(rho).  Hence x_var = b/coef with
    // multiplier (gr_var + rho x_var)/coef (negative ones are left out), and J' n = coef/sqrt(rho) e_var:
    // ordering the columns of J so that the crash variables come first makes R diagonal -- the
    // working set is installed by a row permutation of JT, no reflector needed.
    int* perm = S.piv;
    if (threadIdx.x == 0) {
        for (int j = 0; j < n; ++j) S.t[j] = 0.0;
        int qq = 0;
        for (int i = 0; i < n_crash; ++i) {
            const int c = S.crash[i], var = S.crash[n + i];
            int row; double sgn, b;
            os_get(P, c, row, sgn, b);
            const double coef = row >= 0 ? P.A[(size_t)var * P.nC + row] : 1.0;
            const double xv = b / coef;
            const double uu = (S.gr[var] + rho * xv) / coef;
            if (uu < 0.0) continue;
            S.act[qq] = c; S.u[qq] = uu; S.is_act[c] = 1; S.x[var] = xv;
            perm[qq] = var; S.t[var] = 1.0;
            S.hv[qq] = coef * W.J0T[(size_t)var * n + var];
            ++qq;
        }
        S.ired[5] = qq;
        for (int j = 0; j < n; ++j) if (S.t[j] == 0.0) perm[qq++] = j;
    }
    __syncthreads();
    q = S.ired[5];
    for (int r = wave_id(); r < n; r += QW) {
        const double* src = W.J0T + (size_t)perm[r] * n;
        double* dst = W.JT + (size_t)r * n;
        for (int k = lane_id(); k < n; k += 64) dst[k] = src[k];
    }
    for (int pos = wave_id(); pos < q; pos += QW)
        for (int i = lane_id(); i <= pos; i += 64) {
            W.R[(size_t)pos * n + i] = (i == pos) ? S.hv[pos] : 0.0;
            W.T[(size_t)pos * n + i] = (i == pos) ? 1.0 / S.hv[pos] : 0.0;
        }
    __syncthreads();
    TOC(t_cr, 9);
    }
    int refreshes = 0;
    for (;;) {
        TIC(t_scan);
        // columns to read: all but the variables an active single-entry row or bound holds at exactly zero
        for (int j = threadIdx.x; j < n; j += QT) S.hv[j] = 0.0;
        __syncthreads();
        for (int c = threadIdx.x; c < q; c += QT) {
            int row; double sgn, b;
            os_get(P, S.act[c], row, sgn, b);
            const int var = row < 0 ? -row - 1 : S.svar[row];
            if (var >= 0 && b == 0.0) S.hv[var] = 1.0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int nl = 0;
            for (int j = 0; j < n; ++j) if (S.hv[j] == 0.0) S.live[nl++] = j;
            S.ired[7] = nl;
        }
        __syncthreads();
        rows_times_live(P, S.x, S.ax, S.jend, S.live, S.ired[7]);
        double worst = 0.0; int p = -1;
        for (int c = threadIdx.x; c < P.m1; c += QT) {
            int row; double sgn, b;
            if (S.is_act[c] || !os_get(P, c, row, sgn, b)) continue;
            if (S.is_act[c ^ 1]) {
                // other side of a two-sided row is in the working set: with lb == ub (an equality, e.g.
                // the blocked-move rows) this side holds by construction, whatever rounding says
                const double lo = row >= 0 ? P.lba[row] : P.lbx[-row - 1], hi = row >= 0 ? P.uba[row] : P.ubx[-row - 1];
                if (lo == hi) continue;
            }
            double s = sgn * (row >= 0 ? S.ax[row] : S.x[-row - 1]) - b;
            double tol = 1e-11 * (1.0 + fabs(b));
            if (s < -tol && s < worst) { worst = s; p = c; }
        }
        block_argmin(worst, p, S);
        TOC(t_scan, 0);
        if (p < 0) {
            if (refreshes < 1 && q > 0) {
                // refresh (x,u) from the working set -- guards against drift accumulated over many
                // rank-one steps.  The regularised KKT system is solved through the factors at hand:
                // with J'N' = [R;0] and G^-1 = JJ':  y = R^-T b,  x = J1 y - J2 (J'gr)_2,
                // u = R^-1 (y + (J'gr)_1)   (mathematically the LU solve of that KKT system)
                TIC(t_k0);
                int okr = 1;
                factor_refresh(P, W, S, n, q);
                TOC(t_k0, 7);
                if (okr) {
                    double mn = 0.0;
                    for (int j = threadIdx.x; j < q; j += QT) mn = fmax(mn, -S.up[j]);
                    mn = block_max(mn, S);
                    if (mn > 0.0) okr = 0;
                }
                if (okr) {
                    for (int i = threadIdx.x; i < n; i += QT) S.x[i] = S.xp[i];
                    for (int j = threadIdx.x; j < q; j += QT) S.u[j] = S.up[j];
                    __syncthreads();
                }
                ++refreshes;
                if (okr) continue;
            }
            break;
        }
        if (++iters > max_iter) { status = 2; break; }
        get_normal(P, p, S.np);
        if (threadIdx.x == 0) S.u[q] = 0.0;
        __syncthreads();
        int dropped_guard = 0;
        for (;;) {
            TIC(t_d);
            rowdot(W.JT, n, S.np, S.d);
            TOC(t_d, 1);
            TIC(t_z);
            coldot(W.JT, n, q, S.d, S.z, 1.0);
            TOC(t_z, 2);
            TIC(t_r);
            double pz = 0.0, p1 = 0.0;
            for (int j = threadIdx.x; j < n; j += QT) { double v = S.d[j] * S.d[j]; if (j >= q) pz += v; else p1 += v; }
            double znorm2 = block_sum(pz, S), d1n = block_sum(p1, S);
            // r = R^-1 d1 = T d1 with the explicit inverse (no serial back substitution)
            for (int i = threadIdx.x; i < q; i += QT) {
                double acc = 0.0;
                int k = i;
                for (; k + 8 <= q; k += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = W.T[(size_t)(k + u) * n + i];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += v[u] * S.d[k + u];
                }
                for (; k < q; ++k) acc += W.T[(size_t)k * n + i] * S.d[k];
                S.r[i] = acc;
            }
            __syncthreads();
            TOC(t_r, 3);
            TIC(t_s);
            int z_zero = (znorm2 <= 1e-26 * (1.0 + d1n + znorm2));
            double t1 = INFINITY; int l = -1;
            for (int j = threadIdx.x; j < q; j += QT)
                if (S.r[j] > 0.0) {
                    double t = S.u[j] / S.r[j];
                    if (t < t1) { t1 = t; l = j; }
                }
            block_argmin(t1, l, S);
            if (l < 0) t1 = INFINITY;
            double sp = con_value(P, p, S.x, S);
            double t2 = z_zero ? INFINITY : -sp / znorm2;
            if (t2 < 0.0) t2 = 0.0;
            double t = t1 < t2 ? t1 : t2;
            TOC(t_s, 4);
            if (!isfinite(t)) { status = 1; goto done; }
            if (z_zero || t2 == INFINITY) {
                __syncthreads();
                for (int j = threadIdx.x; j < q; j += QT) S.u[j] -= t * S.r[j];
                if (threadIdx.x == 0) { S.u[q] += t; S.is_act[S.act[l]] = 0; }
                TIC(t_dr0);
                gi_drop(W, S, n, q, l);
                TOC(t_dr0, 6);
                --q;
                if (++dropped_guard > 4 * n + 16) { status = 2; goto done; }
                continue;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += QT) S.x[i] += t * S.z[i];
            for (int j = threadIdx.x; j < q; j += QT) S.u[j] -= t * S.r[j];
            if (threadIdx.x == 0) S.u[q] += t;
            __syncthreads();
            if (t == t2) {
                TIC(t_add);
                if (gi_add(W, S, n, q) == 0) {
                    if (threadIdx.x == 0) { S.act[q] = p; S.is_act[p] = 1; }
                    ++q;
                }
                __syncthreads();
                TOC(t_add, 5);
                break;
            }
            if (threadIdx.x == 0) S.is_act[S.act[l]] = 0;
            TIC(t_dr1);
            gi_drop(W, S, n, q, l);
            TOC(t_dr1, 6);
            --q;
            if (++dropped_guard > 4 * n + 16) { status = 2; goto done; }
        }
    }
done:
    __syncthreads();
    q_out = q;
    iters_out = iters;
    return status;
}

__device__ void carve(Lds& S, unsigned char* base, int n, int nC) {
    double* p = (double*)base;
    const int n2 = 2 * n + 2;
    S.x = p; p += n; S.xc = p; p += n; S.gr = p; p += n; S.np = p; p += n; S.d = p; p += n;
    S.z = p; p += n; S.r = p; p += n; S.u = p; p += n + 2; S.xp = p; p += n; S.up = p; p += n + 2;
    S.t = p; p += n; S.hv = p; p += n;
    S.rhs = p; p += n2; S.sol = p; p += n2; S.res = p; p += n2; S.fcol = p; p += n2;
    S.ax = p; p += nC; S.red = p; p += 8;
    int* ip = (int*)p;
    S.act = ip; ip += n + 2; S.crash = ip; ip += 2 * n; S.piv = ip; ip += n2; S.ired = ip; ip += 8;
    S.jend = ip; ip += nC; S.svar = ip; ip += nC;
    S.kk_kind = ip; ip += n + 2; S.kk_vmap = ip; ip += n; S.kk_free = ip; ip += n; S.kk_gpos = ip; ip += n + 2;
    S.live = ip; ip += n;
    S.is_act = (unsigned char*)ip;
}

__global__ void __launch_bounds__(QT, 2) k_qp_dense(eepacc_qp_args a) {     // two workgroups per CU: 256 VGPRs, no scratch
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = a.nV, nC = a.nC;
    Lds S;
    carve(S, smem, n, nC);
    Ws W;
    {
        double* w = a.ws + (size_t)blockIdx.x * a.ws_stride;
        const size_t nn = (size_t)n * n;
        W.Hs = w; W.J0T = w + nn; W.JT = w + 2 * nn; W.R = w + 3 * nn; W.R2 = w + 4 * nn; W.T = w + 5 * nn; W.T2 = w + 6 * nn; W.K = w + 7 * nn;
    }
    // problems are handed out through a counter: solve times differ by an order of magnitude
    // (proximal rounds of degenerate problems), a static split would leave most workgroups idle
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.ired[6] = atomicAdd(a.counter, 1);
        __syncthreads();
        const int b = S.ired[6];
        if (b >= a.B) break;
        Prob P;
        P.n = n; P.nC = nC; P.m1 = 2 * (nC + n);
        P.H = a.H + (size_t)b * n * n;
        P.g = a.g + (size_t)b * n;
        P.A = a.A + (size_t)b * nC * n;
        P.lba = a.lba ? a.lba + (size_t)b * nC : nullptr;
        P.uba = a.uba ? a.uba + (size_t)b * nC : nullptr;
        P.lbx = a.lbx ? a.lbx + (size_t)b * n : nullptr;
        P.ubx = a.ubx ? a.ubx + (size_t)b * n : nullptr;
        // symmetrised Hessian, diagonal scale
        double hm = 0.0;
        for (int idx = threadIdx.x; idx < n * n; idx += QT) {
            int i = idx / n, j = idx % n;
            W.Hs[idx] = 0.5 * (P.H[idx] + P.H[(size_t)j * n + i]);
            if (i == j) hm = fmax(hm, fabs(P.H[idx]));
        }
        double hmax = block_max(hm, S);
        if (hmax == 0.0) hmax = 1.0;
        double rho = (a.rho_rel > 0.0 ? a.rho_rel : 1e-7) * hmax;
        const int max_prox = a.max_prox > 0 ? a.max_prox : 8;
        double* L = W.K;
        int chol_ok = 0;
        TIC(t_ch);
        // rho = rho0 * 4^k with the smallest k for which H + rho I has a Cholesky factor.  The search
        // starts from the hint k of the previous solve of this problem slot (the FB Hessian needs k = 10
        // every step): try k-1, and if that fails take k, which costs two factorisations instead of k+1;
        // positive definiteness is monotone in rho, so the result is the k of the search from zero.
        const double rho0 = rho;
        int kh = a.rho_k ? a.rho_k[b] : 0;
        if (kh < 0 || kh > 59) kh = 0;
        auto try_k = [&](int k) -> int {
            const double r = rho0 * exp2(2.0 * k);
            for (int idx = threadIdx.x; idx < n * n; idx += QT) {
                int i = idx / n, j = idx % n;
                L[idx] = W.Hs[idx] + (i == j ? r : 0.0);
            }
            __syncthreads();
            const int ok = chol_lower(L, n, S) == 0;
            __syncthreads();
            return ok;
        };
        int kfound = -1, last_ok = -1;
        if (kh > 0) {
            // walk down while the factor exists (it is kept for the smallest such k only if that was the last try)
            int k = kh;
            if (try_k(k - 1)) {
                k = k - 1;
                while (k > 0 && try_k(k - 1)) --k;
                kfound = k;                      // k-1 failed (or k == 0): L holds garbage of the failed try unless k == 0
                last_ok = (k == 0) ? 0 : -1;
            } else {
                while (k < 60 && !try_k(k)) ++k;
                if (k < 60) { kfound = k; last_ok = k; }
            }
        } else {
            int k = 0;
            while (k < 60 && !try_k(k)) ++k;
            if (k < 60) { kfound = k; last_ok = k; }
        }
        if (kfound >= 0 && last_ok != kfound) try_k(kfound);          // re-factor: the last attempt was a failing one
        chol_ok = kfound >= 0;
        if (chol_ok) {
            rho = rho0 * exp2(2.0 * kfound);
            if (a.rho_k && threadIdx.x == 0) a.rho_k[b] = kfound;
        }
        int status = 1, tot_iters = 0, q = 0;
        if (chol_ok) {
            tri_inverse(L, W.J0T, n, S);
            TOC(t_ch, 8);
            // crash list: lower bounds of curvature-free variables with positive cost
            // S.hv[j] = 1 if variable j has any curvature
            for (int j = threadIdx.x; j < n; j += QT) {
                double any = 0.0;
                for (int i = 0; i < n; ++i) any = fmax(any, fmax(fabs(P.H[(size_t)i * n + j]), fabs(P.H[(size_t)j * n + i])));
                S.hv[j] = any;
            }
            // rows with exactly one non-zero: S.ax[i] = column index (or -1)
            for (int i = threadIdx.x; i < nC; i += QT) {
                int nnz = 0, var = -1;
                for (int j = 0; j < n; ++j) if (P.A[(size_t)j * nC + i] != 0.0) { ++nnz; var = j; }
                S.ax[i] = nnz == 1 ? (double)var : -1.0;
                S.jend[i] = var + 1;
                S.svar[i] = nnz == 1 ? var : -1;
            }
            __syncthreads();
            int n_crash = 0;
            if (threadIdx.x == 0) {
                // serial pass keeps the list order of the one-sided list (first bound per variable wins)
                for (int j = 0; j < n; ++j) S.t[j] = 0.0;     // taken flags
                for (int c = 0; c < P.m1 && n_crash < n; c += 2) {   // lower sides only
                    int row; double sgn, bb;
                    if (!os_get(P, c, row, sgn, bb)) continue;
                    int var; double coef;
                    if (row >= 0) {
                        if (S.ax[row] < 0.0) continue;
                        var = (int)S.ax[row];
                        coef = P.A[(size_t)var * nC + row];
                    } else { var = -row - 1; coef = 1.0; }
                    if (coef <= 0.0 || S.t[var] != 0.0 || !(P.g[var] > 0.0) || S.hv[var] != 0.0) continue;
                    S.t[var] = 1.0;
                    S.crash[n_crash] = c;
                    S.crash[n + n_crash] = var;
                    ++n_crash;
                }
                S.ired[4] = n_crash;
            }
            __syncthreads();
            n_crash = S.ired[4];
            for (int i = threadIdx.x; i < n; i += QT) S.xc[i] = a.x0 ? a.x0[(size_t)b * n + i] : 0.0;
            __syncthreads();
            for (int it = 0; it < max_prox; ++it) {
                for (int i = threadIdx.x; i < n; i += QT) S.gr[i] = P.g[i] - rho * S.xc[i];
                __syncthreads();
                int iters = 0;
                TIC(t_gi);
                int rc = gi_solve(P, W, S, rho, n_crash, q, iters, 20 * (n + P.m1) + 100, it > 0);
                TOC(t_gi, 10);
                if (blockIdx.x == 0 && threadIdx.x == 0) { TOC(t_gi, 15); }
                tot_iters += iters;
                if (rc != 0) { status = 1; break; }
                double kkt[3];
                TIC(t_k1);
                int prc = kkt_solve(P, W, S, 0.0, P.g, S.act, q, S.xp, S.up, kkt);
                TOC(t_k1, 7);
#ifdef EEPACC_QP_DEBUG
                if (threadIdx.x == 0 && it == 0) printf("qp %d round %d: q=%d prc=%d kkt=%g %g %g (nr=%d qg=%d)\n", b, it, q, prc, kkt[0], kkt[1], kkt[2], S.ired[2], S.ired[3]);
#endif
                if (prc == 0 && kkt[0] < 1e-9 && kkt[1] < 1e-9 && kkt[2] < 1e-9) {
                    for (int i = threadIdx.x; i < n; i += QT) S.x[i] = S.xp[i];
                    __syncthreads();
                    status = 0;
                    break;
                }
                // Degenerate optimal face (singular KKT matrix, e.g. the FB force split): the proximal
                // rounds only creep along it.  Jump to their limit: KKT solve on the working set with a
                // vanishing proximal term centred at the current point, verified like the exact polish.
                {
                    const double rho2 = 1e-9 * hmax;
                    for (int i = threadIdx.x; i < n; i += QT) S.gr[i] = P.g[i] - rho2 * S.x[i];
                    __syncthreads();
                    double kk2[3];
                    TIC(t_k2);
                    int prc2 = kkt_solve(P, W, S, rho2, S.gr, S.act, q, S.xp, S.up, kk2);
                    TOC(t_k2, 7);
                    if (prc2 == 0 && kk2[0] < 1e-9 && kk2[1] < 1e-9 && kk2[2] < 1e-9) {
                        for (int i = threadIdx.x; i < n; i += QT) S.x[i] = S.xp[i];
                        __syncthreads();
                        status = 0;
                        break;
                    }
                }
                double pdx = 0.0, pnx = 0.0;
                for (int i = threadIdx.x; i < n; i += QT) {
                    double dd = S.x[i] - S.xc[i];
                    pdx += dd * dd; pnx += S.x[i] * S.x[i];
                }
                double dx = block_sum(pdx, S), nx = block_sum(pnx, S);
                for (int i = threadIdx.x; i < n; i += QT) S.xc[i] = S.x[i];
                __syncthreads();
                if (it > 0 && sqrt(dx) <= 1e-13 * (1.0 + sqrt(nx))) { status = 0; break; }
            }
        } else {
            for (int i = threadIdx.x; i < n; i += QT) S.x[i] = 0.0;
        }
        __syncthreads();
        // cost = 1/2 x'Hx + g'x
        double pc = 0.0;
        for (int i = threadIdx.x; i < n; i += QT) {
            double hx = 0.0;
            for (int j = 0; j < n; ++j) hx = fma(P.H[(size_t)i * n + j], S.x[j], hx);
            pc += (0.5 * hx + P.g[i]) * S.x[i];
        }
        double cost = block_sum(pc, S);
        for (int i = threadIdx.x; i < n; i += QT) a.x[(size_t)b * n + i] = S.x[i];
        if (threadIdx.x == 0) {
            if (a.cost) a.cost[b] = cost;
            if (a.status) a.status[b] = status;
            if (a.iters) a.iters[b] = tot_iters;
        }
    }
}

}  // namespace

size_t eepacc_qp_dense_ws_doubles(int nV) {
    const size_t n = (size_t)nV;
    return 7 * n * n + (2 * n + 2) * (2 * n + 2);
}

size_t eepacc_qp_dense_lds_bytes(int nV, int nC) {
    const size_t n = (size_t)nV, n2 = 2 * n + 2;
    size_t dbl = 12 * n + 4 + 4 * n2 + (size_t)nC + 8;
    size_t ints = (n + 2) + 2 * n + n2 + 8 + 2 * (size_t)nC + 5 * n + 4;
    size_t bytes = dbl * 8 + ints * 4 + 2 * ((size_t)nC + n) + 16;
    return (bytes + 15) & ~(size_t)15;
}

#ifdef EEPACC_QP_TIMING
extern "C" int eepacc_debug_qp_prof(long long* out, int reset) {
    long long z[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qp_prof), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_qp_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

hipError_t eepacc_qp_dense_launch(const eepacc_qp_args& a, int grid, hipStream_t stream) {
    size_t lds = eepacc_qp_dense_lds_bytes(a.nV, a.nC);
    hipError_t e = hipFuncSetAttribute((const void*)k_qp_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_qp_dense, dim3(grid), dim3(QT), lds, stream, a);
    return hipGetLastError();
}
