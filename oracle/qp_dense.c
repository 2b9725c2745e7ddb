/*
 * qp_dense.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Dense QP solver standing in for the reference's third-party solver call
 *     sol = QPsolver('h',H,'g',c,'a',G,'lbx',z_lb,'ubx',z_ub,'lba',g_lb,'uba',g_ub)
 * (ABO/RunOpt_ABMPC.m:252, ABO/RunOpt_FBMPC.m:278), i.e. CasADi 3.6.3 `conic` plugin
 * `qpoases` (CAS/include/casadi/config.h:27-34; binary-only Windows DLL, source absent
 * from the reference tree -- SURVEY.md section 8c).  Problem solved, as documented in
 * CAS/+casadi/conic.m:951-966:
 *     min 1/2 x'Hx + g'x   s.t.  lba <= A x <= uba,  lbx <= x <= ubx      (+-inf = absent)
 *
 * qpOASES is an online active-set method; for the convex (AB) QPs the minimiser is unique
 * (SURVEY.md section 8c), so any exact active-set method returns the same point.  This file
 * restates two published algorithms:
 *   (1) Goldfarb & Idnani, "A numerically stable dual method for solving strictly convex
 *       quadratic programs", Math. Prog. 27 (1983): dual active set with the J = L^-T Q /
 *       R factor pair updated by Givens rotations;
 *   (2) the proximal-point regularisation qpOASES itself applies to positive SEMI-definite
 *       Hessians (H + rho I, linear term re-centred at the previous iterate; Ferreau et al.,
 *       "qpOASES: a parametric active-set algorithm for quadratic programming", Math. Prog.
 *       Comp. 6 (2014), section 4.5) -- the AB dense Hessian has 3N zero eigenvalues.
 * After the proximal loop the identified working set is polished by solving the exact
 * (unregularised) KKT system with LU + iterative refinement, and the full KKT conditions
 * are verified; the residuals are returned so tests can assert on them.
 *
 * Parity of this solver with the reference is pinned by the saved-solution goldens
 * (tests/golden/<name>.npz: per-step xi_*, Fm, Fb, a of 871 MPC steps; see tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "oracle.h"

#define IDX(i, j, ld) ((size_t)(i) * (size_t)(ld) + (size_t)(j))

/* ------------------------------------------------------------------------------------ */
/* small dense helpers                                                                   */

/* Cholesky G = L L' (lower, row-major n x n, in place on a copy). returns 0 ok */
static int chol_lower(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double s = A[IDX(j, j, n)];
        for (int k = 0; k < j; ++k) s -= A[IDX(j, k, n)] * A[IDX(j, k, n)];
        if (!(s > 0.0)) return -1;
        double l = sqrt(s);
        A[IDX(j, j, n)] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = A[IDX(i, j, n)];
            for (int k = 0; k < j; ++k) t -= A[IDX(i, k, n)] * A[IDX(j, k, n)];
            A[IDX(i, j, n)] = t / l;
        }
        for (int k = j + 1; k < n; ++k) A[IDX(j, k, n)] = 0.0;
    }
    return 0;
}

/* LU with partial pivoting, in place; piv[n]. returns min |pivot| (0 if singular) */
static double lu_factor(double* A, int n, int* piv) {
    double minpiv = INFINITY;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[IDX(k, k, n)]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(A[IDX(i, k, n)]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (best < minpiv) minpiv = best;
        if (best == 0.0) return 0.0;
        if (p != k)
            for (int j = 0; j < n; ++j) {
                double t = A[IDX(k, j, n)]; A[IDX(k, j, n)] = A[IDX(p, j, n)]; A[IDX(p, j, n)] = t;
            }
        double inv = 1.0 / A[IDX(k, k, n)];
        for (int i = k + 1; i < n; ++i) {
            double f = A[IDX(i, k, n)] * inv;
            if (f != 0.0) {
                A[IDX(i, k, n)] = f;
                double* ri = &A[IDX(i, 0, n)];
                const double* rk = &A[IDX(k, 0, n)];
                for (int j = k + 1; j < n; ++j) ri[j] -= f * rk[j];
            }
        }
    }
    return minpiv;
}

static void lu_solve(const double* LU, const int* piv, int n, double* b) {
    /* rows were swapped whole during factorisation: apply every interchange first */
    for (int k = 0; k < n; ++k) {
        int p = piv[k];
        if (p != k) { double t = b[k]; b[k] = b[p]; b[p] = t; }
    }
    for (int k = 0; k < n; ++k)
        for (int i = k + 1; i < n; ++i) b[i] -= LU[IDX(i, k, n)] * b[k];
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= LU[IDX(i, j, n)] * b[j];
        b[i] = s / LU[IDX(i, i, n)];
    }
}

/* Symmetric eigen-decomposition A = Z diag(d) Z' (Householder tridiagonalisation + implicit QL, the
 * classical EISPACK pair tred2 / tql2; Wilkinson & Reinsch, Handbook for Automatic Computation II,
 * 1971).  a: n x n row-major, overwritten by the eigenvectors (columns); d: eigenvalues.  returns 0 ok. */
static int sym_eig(double* a, int n, double* d) {
    double* e = (double*)calloc(n, sizeof(double));
    for (int i = n - 1; i > 0; --i) {
        int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += fabs(a[IDX(i, k, n)]);
            if (scale == 0.0) e[i] = a[IDX(i, l, n)];
            else {
                for (int k = 0; k <= l; ++k) { a[IDX(i, k, n)] /= scale; h += a[IDX(i, k, n)] * a[IDX(i, k, n)]; }
                double f = a[IDX(i, l, n)];
                double g = f >= 0.0 ? -sqrt(h) : sqrt(h);
                e[i] = scale * g; h -= f * g;
                a[IDX(i, l, n)] = f - g;
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    a[IDX(j, i, n)] = a[IDX(i, j, n)] / h;
                    g = 0.0;
                    for (int k = 0; k <= j; ++k) g += a[IDX(j, k, n)] * a[IDX(i, k, n)];
                    for (int k = j + 1; k <= l; ++k) g += a[IDX(k, j, n)] * a[IDX(i, k, n)];
                    e[j] = g / h;
                    f += e[j] * a[IDX(i, j, n)];
                }
                double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = a[IDX(i, j, n)];
                    e[j] = g = e[j] - hh * f;
                    for (int k = 0; k <= j; ++k) a[IDX(j, k, n)] -= f * e[k] + g * a[IDX(i, k, n)];
                }
            }
        } else e[i] = a[IDX(i, l, n)];
        d[i] = h;
    }
    d[0] = 0.0; e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        int l = i - 1;
        if (d[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g = 0.0;
                for (int k = 0; k <= l; ++k) g += a[IDX(i, k, n)] * a[IDX(k, j, n)];
                for (int k = 0; k <= l; ++k) a[IDX(k, j, n)] -= g * a[IDX(k, i, n)];
            }
        }
        d[i] = a[IDX(i, i, n)];
        a[IDX(i, i, n)] = 1.0;
        for (int j = 0; j <= l; ++j) a[IDX(j, i, n)] = a[IDX(i, j, n)] = 0.0;
    }
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    int rc = 0;
    for (int l = 0; l < n && rc == 0; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                double dd = fabs(d[m]) + fabs(d[m + 1]);
                if (fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 60) { rc = -1; break; }
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? fabs(r) : -fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i], b = c * e[i];
                    e[i + 1] = (r = hypot(f, g));
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    d[i + 1] = g + (p = s * r);
                    g = c * r - b;
                    for (int k = 0; k < n; ++k) {
                        f = a[IDX(k, i + 1, n)];
                        a[IDX(k, i + 1, n)] = s * a[IDX(k, i, n)] + c * f;
                        a[IDX(k, i, n)] = c * a[IDX(k, i, n)] - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    free(e);
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* one-sided constraint list  n_i' x >= b_i  built from the two-sided reference form     */

typedef struct {
    int m;            /* number of one-sided constraints                      */
    int* row;         /* source: >=0 row of A, <0 : -(var+1) simple bound     */
    int* sign;        /* +1: lower side (a'x >= lb) ; -1: upper (-a'x >= -ub) */
    double* b;        /* right-hand side in >= form                           */
} onesided;

static void build_onesided(onesided* C, int nV, int nC, const double* lba, const double* uba,
                           const double* lbx, const double* ubx) {
    int cap = 2 * (nV + nC);
    C->row = (int*)malloc(sizeof(int) * cap);
    C->sign = (int*)malloc(sizeof(int) * cap);
    C->b = (double*)malloc(sizeof(double) * cap);
    int m = 0;
    for (int i = 0; i < nC; ++i) {
        if (lba && isfinite(lba[i])) { C->row[m] = i; C->sign[m] = +1; C->b[m] = lba[i]; ++m; }
        if (uba && isfinite(uba[i])) { C->row[m] = i; C->sign[m] = -1; C->b[m] = -uba[i]; ++m; }
    }
    for (int j = 0; j < nV; ++j) {
        if (lbx && isfinite(lbx[j])) { C->row[m] = -(j + 1); C->sign[m] = +1; C->b[m] = lbx[j]; ++m; }
        if (ubx && isfinite(ubx[j])) { C->row[m] = -(j + 1); C->sign[m] = -1; C->b[m] = -ubx[j]; ++m; }
    }
    C->m = m;
}

static void free_onesided(onesided* C) { free(C->row); free(C->sign); free(C->b); }

/* normal of one-sided constraint i into nrm[nV] */
static void get_normal(const onesided* C, int i, const double* A, int nV, double* nrm) {
    int r = C->row[i];
    double sg = (double)C->sign[i];
    if (r >= 0) {
        const double* a = &A[IDX(r, 0, nV)];
        for (int j = 0; j < nV; ++j) nrm[j] = sg * a[j];
    } else {
        memset(nrm, 0, sizeof(double) * nV);
        nrm[-r - 1] = sg;
    }
}

static double con_value(const onesided* C, int i, const double* A, int nV, const double* x) {
    int r = C->row[i];
    double sg = (double)C->sign[i];
    if (r >= 0) {
        const double* a = &A[IDX(r, 0, nV)];
        double s = 0.0;
        for (int j = 0; j < nV; ++j) s += a[j] * x[j];
        return sg * s - C->b[i];
    }
    return sg * x[-r - 1] - C->b[i];
}

/* ------------------------------------------------------------------------------------ */
/* Goldfarb-Idnani dual active set on a strictly convex QP                               */
/*   min 1/2 x'Gx + g'x  s.t.  n_i'x >= b_i                                              */

typedef struct {
    int n;
    double* J;      /* n x n : L^-T Q                         */
    double* R;      /* n x n upper triangular (q x q used)    */
    double* d;      /* n                                       */
    double* z;      /* n                                       */
    double* r;      /* n                                       */
    double* np;     /* n                                       */
    int* act;       /* active one-sided constraint ids (q)     */
    double* u;      /* multipliers of active set (q+1)         */
    int q;
} gi_work;

static void givens(double a, double b, double* c, double* s) {
    if (b == 0.0) { *c = 1.0; *s = 0.0; return; }
    double h = hypot(a, b);
    *c = a / h; *s = b / h;
}

/* add constraint whose d = J' n has been computed; returns 0 ok, -1 if dependent */
static int gi_add(gi_work* w) {
    int n = w->n, q = w->q;
    double* d = w->d;
    for (int j = n - 1; j > q; --j) {
        double c, s;
        if (d[j] == 0.0) continue;
        givens(d[j - 1], d[j], &c, &s);
        d[j - 1] = c * d[j - 1] + s * d[j];
        d[j] = 0.0;
        for (int k = 0; k < n; ++k) {
            double a = w->J[IDX(k, j - 1, n)], b = w->J[IDX(k, j, n)];
            w->J[IDX(k, j - 1, n)] = c * a + s * b;
            w->J[IDX(k, j, n)] = -s * a + c * b;
        }
    }
    for (int i = 0; i <= q; ++i) w->R[IDX(i, q, n)] = d[i];
    if (fabs(d[q]) <= 1e-14 * (1.0 + fabs(d[0]))) return -1;
    w->q = q + 1;
    return 0;
}

/* drop active constraint at position l */
static void gi_drop(gi_work* w, int l) {
    int n = w->n, q = w->q;
    for (int j = l; j < q - 1; ++j) {
        w->act[j] = w->act[j + 1];
        w->u[j] = w->u[j + 1];
        for (int i = 0; i <= j + 1; ++i) w->R[IDX(i, j, n)] = w->R[IDX(i, j + 1, n)];
    }
    w->u[q - 1] = w->u[q];
    w->u[q] = 0.0;
    w->q = --q;
    for (int j = l; j < q; ++j) {
        double c, s;
        double a = w->R[IDX(j, j, n)], b = w->R[IDX(j + 1, j, n)];
        if (b == 0.0) continue;
        givens(a, b, &c, &s);
        for (int k = j; k < q; ++k) {
            double x = w->R[IDX(j, k, n)], y = w->R[IDX(j + 1, k, n)];
            w->R[IDX(j, k, n)] = c * x + s * y;
            w->R[IDX(j + 1, k, n)] = -s * x + c * y;
        }
        for (int k = 0; k < n; ++k) {
            double x = w->J[IDX(k, j, n)], y = w->J[IDX(k, j + 1, n)];
            w->J[IDX(k, j, n)] = c * x + s * y;
            w->J[IDX(k, j + 1, n)] = -s * x + c * y;
        }
    }
}

static int kkt_polish(int n, const double* H, const double* g, const double* A, const onesided* C,
                      const int* act, int q, double* x, double* u, double kkt[3]);

/* returns 0 solved, 1 infeasible, 2 iteration limit, -1 not PD.
 * crash[0..n_crash): one-sided constraints put into the working set before the first
 * iteration (must give a dual-feasible start; entries with a negative multiplier are
 * discarded).  GI allows any such "solution pair" as a start (Goldfarb & Idnani, sec. 3). */
static int gi_solve(int n, const double* G, const double* g, const double* A, const onesided* C,
                    const int* crash, int n_crash,
                    double* x, int* act_out, double* u_out, int* q_out, int* iters_out,
                    int max_iter) {
    gi_work w;
    w.n = n; w.q = 0;
    w.J = (double*)calloc((size_t)n * n, sizeof(double));
    w.R = (double*)calloc((size_t)n * n, sizeof(double));
    w.d = (double*)calloc(n, sizeof(double));
    w.z = (double*)calloc(n, sizeof(double));
    w.r = (double*)calloc(n, sizeof(double));
    w.np = (double*)calloc(n, sizeof(double));
    w.act = (int*)calloc(n + 1, sizeof(int));
    w.u = (double*)calloc(n + 2, sizeof(double));
    double* L = (double*)malloc(sizeof(double) * (size_t)n * n);
    char* is_act = (char*)calloc(C->m + 1, 1);
    int status = 0, iters = 0;
    memcpy(L, G, sizeof(double) * (size_t)n * n);
    if (chol_lower(L, n) != 0) { status = -1; goto done; }
    /* J = L^-T : solve L' J = I  (column by column) */
    for (int c = 0; c < n; ++c) {
        for (int i = n - 1; i >= 0; --i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k < n; ++k) s -= L[IDX(k, i, n)] * w.J[IDX(k, c, n)];
            w.J[IDX(i, c, n)] = s / L[IDX(i, i, n)];
        }
    }
    /* x = -G^-1 g = -J J' g */
    for (int j = 0; j < n; ++j) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += w.J[IDX(k, j, n)] * g[k];
        w.d[j] = s;
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += w.J[IDX(i, j, n)] * w.d[j];
        x[i] = -s;
    }
    if (n_crash > 0) {
        int* cw = (int*)malloc(sizeof(int) * (n_crash + 1));
        double* cu = (double*)malloc(sizeof(double) * (n_crash + 1));
        double* cx = (double*)malloc(sizeof(double) * n);
        double kk[3];
        int nc = 0;
        for (int i = 0; i < n_crash && i < n; ++i) cw[nc++] = crash[i];
        for (int pass = 0; pass < 4 && nc > 0; ++pass) {
            if (kkt_polish(n, G, g, A, C, cw, nc, cx, cu, kk) != 0) { nc = 0; break; }
            int m2 = 0, dropped = 0;
            for (int i = 0; i < nc; ++i) {
                if (cu[i] < 0.0) { ++dropped; continue; }
                cw[m2] = cw[i]; cu[m2] = cu[i]; ++m2;
            }
            nc = m2;
            if (!dropped) break;
            if (pass == 3) nc = 0;
        }
        for (int i = 0; i < nc; ++i) {
            get_normal(C, cw[i], A, n, w.np);
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += w.J[IDX(k, j, n)] * w.np[k];
                w.d[j] = s;
            }
            if (gi_add(&w) == 0) {
                w.act[w.q - 1] = cw[i];
                w.u[w.q - 1] = cu[i];
                is_act[cw[i]] = 1;
            } else { nc = -1; break; }
        }
        if (nc > 0) memcpy(x, cx, sizeof(double) * n);
        if (nc < 0) {   /* dependent crash set: fall back to the cold start */
            for (int i = 0; i < C->m; ++i) is_act[i] = 0;
            free(cw); free(cu); free(cx);
            status = gi_solve(n, G, g, A, C, NULL, 0, x, act_out, u_out, q_out, iters_out, max_iter);
            free(w.J); free(w.R); free(w.d); free(w.z); free(w.r); free(w.np); free(w.act); free(w.u);
            free(L); free(is_act);
            return status;
        }
        free(cw); free(cu); free(cx);
    }
    int refreshes = 0;
    for (;;) {
        /* most violated inactive constraint (scaled by row magnitude) */
        int p = -1;
        double worst = 0.0;
        for (int i = 0; i < C->m; ++i) {
            if (is_act[i]) continue;
            double s = con_value(C, i, A, n, x);
            double tol = 1e-11 * (1.0 + fabs(C->b[i]));
            if (s < -tol && s < worst) { worst = s; p = i; }
        }
        if (p < 0) {
            /* refresh (x,u) from the working set by a direct KKT solve and look again: guards
             * against drift accumulated over many rank-one steps */
            if (refreshes < 3 && w.q > 0) {
                double kk[3];
                double* cx = (double*)malloc(sizeof(double) * n);
                double* cu = (double*)malloc(sizeof(double) * (w.q + 1));
                int okr = kkt_polish(n, G, g, A, C, w.act, w.q, cx, cu, kk) == 0;
                if (okr) for (int j = 0; j < w.q; ++j) if (cu[j] < 0.0) okr = 0;
                if (okr) {
                    memcpy(x, cx, sizeof(double) * n);
                    memcpy(w.u, cu, sizeof(double) * w.q);
                }
                free(cx); free(cu);
                ++refreshes;
                if (okr) continue;
            }
            break;
        }
        if (++iters > max_iter) { status = 2; break; }
        get_normal(C, p, A, n, w.np);
        w.u[w.q] = 0.0;
        int dropped_guard = 0;
        for (;;) {
            int q = w.q;
            /* d = J' np */
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += w.J[IDX(k, j, n)] * w.np[k];
                w.d[j] = s;
            }
            /* z = J2 d2 */
            double znorm2 = 0.0;
            for (int j = q; j < n; ++j) znorm2 += w.d[j] * w.d[j];
            for (int i = 0; i < n; ++i) {
                double s = 0.0;
                for (int j = q; j < n; ++j) s += w.J[IDX(i, j, n)] * w.d[j];
                w.z[i] = s;
            }
            /* r = R^-1 d1 */
            for (int i = q - 1; i >= 0; --i) {
                double s = w.d[i];
                for (int k = i + 1; k < q; ++k) s -= w.R[IDX(i, k, n)] * w.r[k];
                w.r[i] = s / w.R[IDX(i, i, n)];
            }
            double d1n = 0.0;
            for (int j = 0; j < q; ++j) d1n += w.d[j] * w.d[j];
            int z_zero = (znorm2 <= 1e-26 * (1.0 + d1n + znorm2));
            /* dual step length */
            double t1 = INFINITY; int l = -1;
            for (int j = 0; j < q; ++j)
                if (w.r[j] > 0.0) {
                    double t = w.u[j] / w.r[j];
                    if (t < t1) { t1 = t; l = j; }
                }
            double sp = con_value(C, p, A, n, x);
            double t2 = z_zero ? INFINITY : -sp / znorm2;   /* z'np = |d2|^2 */
            if (t2 < 0.0) t2 = 0.0;
            double t = t1 < t2 ? t1 : t2;
            if (!isfinite(t)) { status = 1; goto done; }
            if (z_zero || t2 == INFINITY) {
                for (int j = 0; j < q; ++j) w.u[j] -= t * w.r[j];
                w.u[q] += t;
                is_act[w.act[l]] = 0;
                gi_drop(&w, l);
                if (++dropped_guard > 4 * n + 16) { status = 2; goto done; }
                continue;
            }
            for (int i = 0; i < n; ++i) x[i] += t * w.z[i];
            for (int j = 0; j < q; ++j) w.u[j] -= t * w.r[j];
            w.u[q] += t;
            if (t == t2) {
                /* full step: add p (d still valid because J unchanged since computed) */
                if (gi_add(&w) != 0) {
                    /* numerically dependent: treat as satisfied, do not add */
                    is_act[p] = 0;
                } else {
                    w.act[w.q - 1] = p;
                    is_act[p] = 1;
                }
                break;
            }
            is_act[w.act[l]] = 0;
            gi_drop(&w, l);
            if (++dropped_guard > 4 * n + 16) { status = 2; goto done; }
        }
    }
done:
    *q_out = w.q;
    for (int j = 0; j < w.q; ++j) { act_out[j] = w.act[j]; u_out[j] = w.u[j]; }
    *iters_out = iters;
    free(w.J); free(w.R); free(w.d); free(w.z); free(w.r); free(w.np); free(w.act); free(w.u);
    free(L); free(is_act);
    return status;
}

/* ------------------------------------------------------------------------------------ */
/* exact KKT polish on a fixed working set + verification                                */

/* Solves [H N'; N 0][x; -u] = [-g; b] for the working set `act` (q one-sided constraints,
 * n_i'x = b_i).  Returns 0 if the system was solved; fills kkt[3]: stationarity inf-norm,
 * worst primal violation over ALL constraints, most negative multiplier. */
static int kkt_polish(int n, const double* H, const double* g, const double* A, const onesided* C,
                      const int* act, int q, double* x, double* u, double kkt[3]) {
    int N = n + q;
    double* K = (double*)calloc((size_t)N * N, sizeof(double));
    double* K0 = (double*)malloc(sizeof(double) * (size_t)N * N);
    double* rhs = (double*)calloc(N, sizeof(double));
    double* sol = (double*)calloc(N, sizeof(double));
    double* res = (double*)calloc(N, sizeof(double));
    double* nrm = (double*)malloc(sizeof(double) * n);
    int* piv = (int*)malloc(sizeof(int) * N);
    int rc = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) K[IDX(i, j, N)] = 0.5 * (H[IDX(i, j, n)] + H[IDX(j, i, n)]);
    for (int c = 0; c < q; ++c) {
        get_normal(C, act[c], A, n, nrm);
        for (int j = 0; j < n; ++j) { K[IDX(n + c, j, N)] = nrm[j]; K[IDX(j, n + c, N)] = nrm[j]; }
        rhs[n + c] = C->b[act[c]];
    }
    for (int i = 0; i < n; ++i) rhs[i] = -g[i];
    memcpy(K0, K, sizeof(double) * (size_t)N * N);
    double minpiv = lu_factor(K, N, piv);
    if (minpiv < 1e-13) { rc = -1; goto out; }
    memcpy(sol, rhs, sizeof(double) * N);
    lu_solve(K, piv, N, sol);
    for (int it = 0; it < 3; ++it) {            /* iterative refinement */
        for (int i = 0; i < N; ++i) {
            long double s = rhs[i];
            for (int j = 0; j < N; ++j) s -= (long double)K0[IDX(i, j, N)] * sol[j];
            res[i] = (double)s;
        }
        lu_solve(K, piv, N, res);
        for (int i = 0; i < N; ++i) sol[i] += res[i];
    }
    for (int i = 0; i < n; ++i) x[i] = sol[i];
    for (int c = 0; c < q; ++c) u[c] = -sol[n + c];
    /* verification */
    {
        double stat = 0.0, pviol = 0.0, dneg = 0.0;
        for (int i = 0; i < n; ++i) {
            long double s = g[i];
            for (int j = 0; j < n; ++j) s += (long double)K0[IDX(i, j, N)] * x[j];
            for (int c = 0; c < q; ++c) s -= (long double)K0[IDX(i, n + c, N)] * u[c];
            if (fabs((double)s) > stat) stat = fabs((double)s);
        }
        for (int i = 0; i < C->m; ++i) {
            double v = con_value(C, i, A, n, x) / (1.0 + fabs(C->b[i]));
            if (-v > pviol) pviol = -v;
        }
        double scale = 1.0;
        for (int i = 0; i < n; ++i) if (fabs(g[i]) > scale) scale = fabs(g[i]);
        for (int c = 0; c < q; ++c) {
            if (-u[c] > dneg) dneg = -u[c];
            if (fabs(u[c]) > scale) scale = fabs(u[c]);
        }
        /* stationarity and dual residuals are reported relative to the gradient scale */
        kkt[0] = stat / scale; kkt[1] = pviol; kkt[2] = dneg / scale;
    }
out:
    free(K); free(K0); free(rhs); free(sol); free(res); free(nrm); free(piv);
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* public entry: conic('qpoases') stand-in                                               */

int orc_qp_solve_dense(int nV, int nC, const double* H, const double* g, const double* A,
                       const double* lba, const double* uba, const double* lbx, const double* ubx,
                       const double* x0, double rho_rel, int max_prox,
                       double* x, double* cost, orc_qp_stats* st) {
    onesided C;
    build_onesided(&C, nV, nC, lba, uba, lbx, ubx);
    double* Gr = (double*)malloc(sizeof(double) * (size_t)nV * nV);
    double* gr = (double*)malloc(sizeof(double) * nV);
    double* xc = (double*)calloc(nV, sizeof(double));
    double* xp = (double*)malloc(sizeof(double) * nV);
    double* up = (double*)malloc(sizeof(double) * (nV + 2));
    int* act = (int*)malloc(sizeof(int) * (nV + 2));
    double* u = (double*)malloc(sizeof(double) * (nV + 2));
    int q = 0, status = 1, tot_iters = 0, prox_used = 0, polished = 0;
    double kkt[3] = {INFINITY, INFINITY, INFINITY};
    double hmax = 0.0;
    for (int i = 0; i < nV; ++i) if (fabs(H[IDX(i, i, nV)]) > hmax) hmax = fabs(H[IDX(i, i, nV)]);
    if (hmax == 0.0) hmax = 1.0;
    /* Two regularisations of a Hessian that is not positive definite:
     *  rho_rel >= 0 : H + rho I (uniform; what qpOASES does for positive SEMI-definite Hessians).  Enough for
     *                 the AB QPs; on the indefinite FB Hessian rho has to exceed |lambda_min| ~ 0.1 and the
     *                 proximal rounds then creep in every direction whose curvature is small against rho.
     *  rho_rel <  0 : spectral modification H + M, M = Z diag(max(0, delta - lambda_i)) Z' with
     *                 delta = |rho_rel| max|lambda|: eigenvalues below delta are lifted to delta, everything
     *                 else is left alone, so the rounds are exact Newton steps in the curved directions and
     *                 long steps (cost / delta, stopped by the constraints) in the flat or negative ones.  The
     *                 fixed point is a KKT point of the unmodified problem (M (x - xc) vanishes there) and the
     *                 exact KKT polish / verification below is the same for both. */
    const int spectral = rho_rel < 0.0;
    int gi_failed = 0;
    double* Mreg = NULL;
    if (rho_rel == 0.0) rho_rel = 1e-7;
    if (max_prox <= 0) max_prox = spectral ? 60 : 8;
    double rho = fabs(rho_rel) * hmax;
    if (spectral) {
        double* Z = (double*)malloc(sizeof(double) * (size_t)nV * nV);
        double* ev = (double*)malloc(sizeof(double) * nV);
        Mreg = (double*)calloc((size_t)nV * nV, sizeof(double));
        for (int i = 0; i < nV; ++i)
            for (int j = 0; j < nV; ++j) Z[IDX(i, j, nV)] = 0.5 * (H[IDX(i, j, nV)] + H[IDX(j, i, nV)]);
        if (sym_eig(Z, nV, ev) != 0) { free(Z); free(ev); free(Mreg); Mreg = NULL; }
        else {
            double emax = 0.0;
            for (int i = 0; i < nV; ++i) if (fabs(ev[i]) > emax) emax = fabs(ev[i]);
            if (emax == 0.0) emax = 1.0;
            rho = fabs(rho_rel) * emax;
            for (int k = 0; k < nV; ++k) {
                const double add = rho - ev[k];
                if (!(add > 0.0)) continue;
                for (int i = 0; i < nV; ++i) {
                    const double zi = add * Z[IDX(i, k, nV)];
                    if (zi == 0.0) continue;
                    for (int j = 0; j < nV; ++j) Mreg[IDX(i, j, nV)] += zi * Z[IDX(j, k, nV)];
                }
            }
            for (int i = 0; i < nV; ++i)
                for (int j = 0; j < nV; ++j)
                    Gr[IDX(i, j, nV)] = 0.5 * (H[IDX(i, j, nV)] + H[IDX(j, i, nV)]) + 0.5 * (Mreg[IDX(i, j, nV)] + Mreg[IDX(j, i, nV)]);
            free(Z); free(ev);
        }
    }
    /* uniform mode: an indefinite H needs rho above its most negative eigenvalue: bump until Cholesky of
     * H + rho I succeeds */
    for (int tries = 0; tries < 60 && !Mreg; ++tries) {
        for (int i = 0; i < nV; ++i)
            for (int j = 0; j < nV; ++j)
                Gr[IDX(i, j, nV)] = 0.5 * (H[IDX(i, j, nV)] + H[IDX(j, i, nV)]) + (i == j ? rho : 0.0);
        double* T = (double*)malloc(sizeof(double) * (size_t)nV * nV);
        memcpy(T, Gr, sizeof(double) * (size_t)nV * nV);
        int ok = chol_lower(T, nV) == 0;
        free(T);
        if (ok) break;
        rho *= 4.0;
    }
    if (x0) memcpy(xc, x0, sizeof(double) * nV);
    /* crash working set: lower bounds of variables without curvature and with positive
     * cost (the slack bounds xi >= 0, which the reference writes as rows of A,
     * ABO/.../CreateQP_AB.m:264-279): at the minimiser of the regularised problem restricted
     * to them every such variable sits on its bound with multiplier g_j > 0. */
    int* crash = (int*)malloc(sizeof(int) * (C.m + 1));
    int n_crash = 0;
    {
        char* taken = (char*)calloc(nV, 1);
        for (int i = 0; i < C.m && n_crash < nV; ++i) {
            if (C.sign[i] != +1) continue;
            int var = -1; double coef = 0.0;
            if (C.row[i] >= 0) {
                const double* a = &A[IDX(C.row[i], 0, nV)];
                int nnz = 0;
                for (int j = 0; j < nV; ++j) if (a[j] != 0.0) { ++nnz; var = j; coef = a[j]; }
                if (nnz != 1) continue;
            } else { var = -C.row[i] - 1; coef = 1.0; }
            if (coef <= 0.0 || taken[var] || !(g[var] > 0.0)) continue;
            int flat = 1;
            for (int j = 0; j < nV; ++j) if (H[IDX(var, j, nV)] != 0.0 || H[IDX(j, var, nV)] != 0.0) { flat = 0; break; }
            if (!flat) continue;
            taken[var] = 1;
            crash[n_crash++] = i;
        }
        free(taken);
    }
    for (int it = 0; it < max_prox; ++it) {
        if (Mreg) {
            for (int i = 0; i < nV; ++i) {
                double sacc = g[i];
                for (int j = 0; j < nV; ++j) sacc -= 0.5 * (Mreg[IDX(i, j, nV)] + Mreg[IDX(j, i, nV)]) * xc[j];
                gr[i] = sacc;
            }
        } else {
            for (int i = 0; i < nV; ++i) gr[i] = g[i] - rho * xc[i];
        }
        int iters = 0;
        /* spectral mode: later rounds start from the previous round's working set (entries whose multiplier
         * turns negative for the new linear term are discarded by gi_solve) */
        const int warm_ws = Mreg && it > 0 && q > 0;
        if (warm_ws) memcpy(crash, act, sizeof(int) * q);
        int rc = gi_solve(nV, Gr, gr, A, &C, crash, warm_ws ? q : n_crash, x, act, u, &q, &iters, 20 * (nV + C.m) + 100);
        tot_iters += iters;
        prox_used = it + 1;
        if (rc != 0) { status = 1; gi_failed = 1; break; }
        /* exact polish on the identified working set */
        memcpy(xp, x, sizeof(double) * nV);
        int prc = kkt_polish(nV, H, g, A, &C, act, q, xp, up, kkt);
        if (getenv("ORC_DEBUG")) {
            fprintf(stderr, "prox %d: gi rc=%d iters=%d q=%d polish rc=%d kkt=%g %g %g\n", it, rc, iters, q, prc, kkt[0], kkt[1], kkt[2]);
            for (int c2 = 0; c2 < q; ++c2) fprintf(stderr, " %d%c(%.3g)", C.row[act[c2]], C.sign[act[c2]] > 0 ? 'L' : 'U', u[c2]);
            fprintf(stderr, "\n");
        }
        if (prc == 0 && kkt[0] < 1e-9 && kkt[1] < 1e-9 && kkt[2] < 1e-9) {
            memcpy(x, xp, sizeof(double) * nV);
            memcpy(u, up, sizeof(double) * q);
            status = 0; polished = 1;
            break;
        }
        /* Degenerate optimal face (singular KKT matrix, e.g. the FB force split): the
         * proximal rounds then only creep along that face.  Jump to their limit: KKT solve on the
         * working set with a vanishing proximal term centred at the current point, verified like the
         * exact polish (a wrong working set fails the verification and the rounds go on). */
        {
            const int same = 1;
            if (same) {
                const double rho2 = 1e-9 * hmax;
                double* Hr = (double*)malloc(sizeof(double) * (size_t)nV * nV);
                double* gr2 = (double*)malloc(sizeof(double) * nV);
                for (int i = 0; i < nV; ++i) {
                    for (int j = 0; j < nV; ++j)
                        Hr[IDX(i, j, nV)] = 0.5 * (H[IDX(i, j, nV)] + H[IDX(j, i, nV)]) + (i == j ? rho2 : 0.0);
                    gr2[i] = g[i] - rho2 * x[i];
                }
                double kk2[3];
                int prc2 = kkt_polish(nV, Hr, gr2, A, &C, act, q, xp, up, kk2);
                free(Hr); free(gr2);
                if (getenv("ORC_DEBUG")) fprintf(stderr, "   face solve rc=%d kkt=%g %g %g\n", prc2, kk2[0], kk2[1], kk2[2]);
                if (prc2 == 0 && kk2[0] < 1e-9 && kk2[1] < 1e-9 && kk2[2] < 1e-9) {
                    memcpy(x, xp, sizeof(double) * nV);
                    memcpy(u, up, sizeof(double) * q);
                    kkt[0] = kk2[0]; kkt[1] = kk2[1]; kkt[2] = kk2[2];
                    status = 0; polished = 2;
                    break;
                }
            }
        }
        /* converged proximal sequence without a clean polish (degenerate vertex): accept */
        double dx = 0.0, nx = 0.0;
        for (int i = 0; i < nV; ++i) { dx += (x[i] - xc[i]) * (x[i] - xc[i]); nx += x[i] * x[i]; }
        memcpy(xc, x, sizeof(double) * nV);
        if (getenv("ORC_DEBUG")) fprintf(stderr, "   prox step |dx| = %g, |x| = %g, rho = %g\n", sqrt(dx), sqrt(nx), rho);
        if (it > 0 && sqrt(dx) <= 1e-13 * (1.0 + sqrt(nx))) { status = 0; break; }
    }
    if (cost) {
        long double s = 0.0L;
        for (int i = 0; i < nV; ++i) {
            long double hx = 0.0L;
            for (int j = 0; j < nV; ++j) hx += (long double)H[IDX(i, j, nV)] * x[j];
            s += (0.5L * hx + g[i]) * x[i];
        }
        *cost = (double)s;
    }
    if (st) {
        st->status = status; st->iterations = tot_iters; st->prox_iterations = prox_used;
        st->n_active = q; st->polished = polished;
        st->kkt_stationarity = kkt[0]; st->kkt_primal = kkt[1]; st->kkt_dual = kkt[2];
        st->rho = rho;
    }
    free(crash);
    free(Mreg);
    free_onesided(&C);
    free(Gr); free(gr); free(xc); free(xp); free(up); free(act); free(u);
    /* spectral mode: a floor of 1e-8 max|lambda| leaves the regularised Hessian with a condition number of 1e8, at
     * which the dual active set can misjudge a linear dependence among many active rows (seen with the equality rows
     * of blocked moves).  A failed inner solve is repeated with a hundred times larger floor (more, shorter rounds). */
    if (status != 0 && gi_failed && spectral && fabs(rho_rel) < 1e-4)
        return orc_qp_solve_dense(nV, nC, H, g, A, lba, uba, lbx, ubx, x0, rho_rel * 100.0, max_prox, x, cost, st);
    return status;
}
