/*
 * eepacc_nlp.h -- C ABI of the batched function evaluator of the full-route problem of RunOpt_NLP.
 *
 * ABO/RunOpt_NLP.m (ORIG/RunOpt_NLP.m is the same file) builds one nonlinear program per route
 * (multiple shooting, :296-501: per interval k the controls U_k = [Fm,Fb,xi_v,xi_h,xi_s,xi_f] and the
 * node X_{k+1} = [s,v,theta,j]) and hands it to IPOPT (:505-510).  What IPOPT calls back into, once per
 * iteration, are CasADi's generated functions nlp_f, nlp_g, nlp_grad_f and nlp_jac_g; everything in them is
 * stage-local, so for a batch of routes it is one thread per (route, interval).  eepacc_nlp_eval is that
 * evaluation for B routes at once: the objective, every constraint row, the objective gradient and the
 * Jacobian blocks of the RK4 x 4 integrator (the only rows whose derivatives are not closed-form).
 *
 * SURVEY.md section 8f rank 2 asks for the whole of RunOpt_NLP as a batched solver on the GPU.  This header declares the
 * operators of a structured interior-point iteration: the function evaluation (eepacc_nlp_eval, in the reference's own
 * variable layout), the Newton-system assembly in stage form (eepacc_nlp_newton), its stage-wise factorisation
 * (eepacc_nlp_riccati), the closed-loop nonlinear forward pass (eepacc_nlp_rollout) and the rows along a step
 * (eepacc_nlp_rowdir).  The per-route scalar logic on top (barrier parameter, Levenberg term, step lengths, accept /
 * reject) is host code: eepacc_mpc_casadi_matlab_amd/nlp.py (NlpSolver, RunOpt_NLP); DESIGN.md section 3.8.
 *
 * Layout: batch-major structure-of-arrays like eepacc.h -- item (k, c) of route i is x[(k*C + c)*B + i].
 * (eepacc_nlp_eval); the stage-form operators are route-major (a wavefront or a thread streams its own route), as their
 * comments say.  All pointers named *_dev are device allocations on the current device; tables in eepacc_nlp_problem are
 * host pointers that eepacc_nlp_create copies.
 */
#ifndef EEPACC_NLP_H
#define EEPACC_NLP_H

#include <stdint.h>
#include "eepacc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EEPACC_NLP_MAX_KNOTS 64    /* per lookup table */
#define EEPACC_NLP_MAX_TL     8    /* traffic lights   */
#define EEPACC_NLP_NX         4    /* s, v, theta, j            (RunOpt_NLP.m:204-209) */
#define EEPACC_NLP_NU         6    /* Fm, Fb, xi_v, xi_h, xi_s, xi_f  (:212-222)       */

/* One route family: everything RunOpt_NLP.m:17-184 unpacks or precomputes on the host.  Lookup tables are the
 * knots / values handed to casadi.interpolant('LUT','linear',...) (linear interpolation, linear extrapolation
 * from the end segments). */
typedef struct eepacc_nlp_problem {
    int32_t N;                       /* t_sim / Ts (:189)                                             */
    int32_t n_tl;                    /* traffic lights (:120-157); 0 = none                           */
    int32_t flat;                    /* sum(slope) < 1e-1: theta rows are  theta_{k+1} = 0 (:363-364)  */
    int32_t pad;
    double  Ts;
    double  W[7];                    /* W_NLP = [w_P,w_a,w_j,w_v,w_h,w_s,w_f] (:53-61)                 */
    double  b[21];                   /* power fit of the objective: b_fifthOrder, or b_quadr followed by 15 zeros
                                        (useFifthOrderFit_NLP, :226-236)                              */
    double  s_goal, h_min, tau_min, alpha_TTL;
    int32_t n_vlim, n_curv, n_slope, n_stop, n_vinc, pad2;
    const double *s_vlim, *v_vlim;   /* speedLimLookup (:69)                                          */
    const double *s_curv, *curvature;/* curvatureLookup (:73)                                         */
    const double *s_slope, *slope;   /* slopeLookup (:65)                                             */
    const double *s_stop, *v_stop;   /* stopMaxVelLookup (:88-117)                                    */
    const double *s_vinc, *v_vinc;   /* velIncentiveLookup (:160-184)                                 */
    const double *tl_s;              /* [n_tl][3] knots of tlMaxVelLookup{i} (:141-142)               */
    double        tl_v[3];           /* its values [stopRefvelIncr, TLstopVel, stopRefvelIncr]        */
    const double *tl_state;          /* [n_tl][N] tlSPATLookup{i}(k*Ts): 0.2 red / 1e3 green (:144-154)*/
} eepacc_nlp_problem;

typedef struct eepacc_nlp_handle eepacc_nlp_handle;

/* Rows per interval in `ineq` (orientation: value <= 0), in the order of RunOpt_NLP.m:378-501 followed by the
 * variable bounds of :330-333,352-355:  17 + 2*n_tl + 10 (+1 when s_goal is finite). */
int eepacc_nlp_rows(const eepacc_nlp_problem* p);
int eepacc_nlp_sizeof_problem(void);   /* sizeof(eepacc_nlp_problem) as compiled: binding self-check */

/* Replaces the problem construction of RunOpt_NLP.m:186-503 (tables to the device). */
int  eepacc_nlp_create(eepacc_nlp_handle** out, const eepacc_nlp_problem* p, const eepacc_vehicle* V, int device);
void eepacc_nlp_destroy(eepacc_nlp_handle* h);

/* nlp_f / nlp_g / nlp_grad_f / the integrator blocks of nlp_jac_g for B routes (RunOpt_NLP.m:505-510: what
 * `solver(...)` evaluates every iteration).
 *   s_tv_dev [N][B]          lead-vehicle position, sample k is the one interval k reads (s_tv(k+1), :488-499)
 *   X_dev    [N+1][4][B]     nodes (s, v, theta, j); node 0 is x_init (:210)
 *   U_dev    [N][6][B]       controls and slacks
 *   J_dev    [B]             objective (:343 accumulated)
 *   eq_dev   [N][4][B]       continuity of s and v (RK4 end state - next node), theta row, jerk row (:357-376)
 *   ineq_dev [N][R][B]       R = eepacc_nlp_rows(): every inequality row and bound as  value <= 0
 *   gradJ_dev [N][10][B]     dJ / d(s_k, v_k, theta_k, j_k, U_k)   (may be NULL)
 *   jacF_dev  [N][2][3][B]   d(s_end, v_end) / d(v_k, theta_k, Fm_k + Fb_k); d s_end / d s_k = 1  (may be NULL)
 * `stream` is a hipStream_t (NULL = default stream).  Asynchronous; eepacc_nlp_synchronize waits. */
int eepacc_nlp_eval(eepacc_nlp_handle* h, int B, const double* s_tv_dev, const double* X_dev, const double* U_dev,
                    double* J_dev, double* eq_dev, double* ineq_dev, double* gradJ_dev, double* jacF_dev,
                    void* stream);
int eepacc_nlp_synchronize(eepacc_nlp_handle* h, void* stream);

/* Newton-system assembly of one interior-point iteration in stage form (chi = (s, v, p, j): p_k is the acceleration at
 * node k under the previous force -- the bracket of the jerk row, RunOpt_NLP.m:371-375 --, theta substituted): for every
 * (route, interval) the exact Lagrangian Hessian (second-order sensitivities of the RK4 x 4 integrator and of the
 * running cost, row curvature), the barrier terms of all rows and the linearised dynamics, condensed to the blocks
 * eepacc_nlp_riccati takes.  Route-major arrays: chi [B][N+1][4], u [B][N][6], lam / t [B][N][R] (multipliers and slacks of
 * the rows, R = eepacc_nlp_rows), nu [B][N+1][4] (costates), s_tv [B][N]; mu [B] = barrier parameter of each route, sigma =
 * objective scale.
 * Outputs Q [B][N][10][10], q [B][N][10], AB [B][N][4][10], c [B][N][4] and, if not NULL, the row values [B][N][R] and
 * qlam [B][N][10], the stage gradient of the Lagrangian with the multipliers lam (eepacc_nlp_riccati turns it into the dual
 * residual). */
int eepacc_nlp_newton(eepacc_nlp_handle* h, int B, const double* mu_dev, double sigma, const double* s_tv_dev, const double* chi_dev,
                      const double* u_dev, const double* lam_dev, const double* t_dev, const double* nu_dev,
                      double* Q_dev, double* q_dev, double* AB_dev, double* c_dev, double* rows_dev, double* qlam_dev,
                      void* stream);

/* Stage-wise factorisation of one Newton system per route: the linear solve inside every interior-point iteration
 * (what IPOPT hands to MUMPS in the reference, RunOpt_NLP.m:505-510), exploiting that the problem is an optimal-control
 * problem: with chi = (s, v, p, j), u = the six controls / slacks and w_k = (chi_k, u_k) it solves
 *     min  sum_k  1/2 w_k' Q_k w_k + q_k' w_k     s.t.  dchi_{k+1} = AB_k w_k + c_k,  dchi_0 = 0
 * by the Riccati recursion (backward sweep over the N stages, forward sweep for the step and the costates).  The sweep
 * is the only serial part of an iteration: one wavefront per route walks the stages, routes run side by side.
 *   Q_dev  [B][N][10][10]  stage Hessians (symmetric, both triangles), q_dev [B][N][10], AB_dev [B][N][4][10],
 *   c_dev  [B][N][4]       (route-major: a wavefront streams its own route's stages)
 *   reg_dev [B]            Levenberg term of the route; reg * reg_scale[i] is added to the diagonal of control i
 *   dchi_dev [B][N+1][4], du_dev [B][N][6], nu_dev [B][N+1][4]   step and costates (nu_0 and nu_N are 0)
 *   work_dev [B][N][50]    gains and value-function blocks between the two sweeps
 *   status_dev [B]         0, or 1 + the first stage (counted from the end) whose control block is not positive definite
 *                          (pivot <= 1e-10 x its diagonal entry): the caller raises reg and calls again
 *   gnorm_dev [B]          (may be NULL) largest |reduced gradient| over the stages and controls: the dual residual of the
 *                          barrier problem at the point the system was assembled at; with qlam_dev (eepacc_nlp_newton)
 *                          the one of the Lagrangian with the current multipliers (adjoint recursion along the sweep) */
int eepacc_nlp_riccati(int device, int B, int N, const double* Q_dev, const double* q_dev, const double* AB_dev,
                       const double* c_dev, const double* reg_dev, const double reg_scale[6], double* dchi_dev,
                       double* du_dev, double* nu_dev, double* work_dev, int32_t* status_dev, double* gnorm_dev,
                       const double* qlam_dev, void* stream);

/* Row values r [B][N][R] at (chi, u) and, if jdy_dev is not NULL, their directional derivative Jr (dchi_{k+1}, du_k) along a
 * step: what the step-length rules and the multiplier update of an interior-point iteration need. */
int eepacc_nlp_rowdir(eepacc_nlp_handle* h, int B, const double* s_tv_dev, const double* chi_dev, const double* u_dev,
                      const double* dchi_dev, const double* du_dev, double* rows_dev, double* jdy_dev, void* stream);


/* Closed-loop nonlinear forward pass of a step (the dynamics hold at every iterate of the solver): for every route
 *   u_k = u_k + alpha * kf_k + K_k (chi_k_new - chi_k),   chi_{k+1}_new = f(chi_k_new, u_k)   (RK4 x 4, RunOpt_NLP.m:262-278)
 * with the gains K | kf that eepacc_nlp_riccati left in its work array [B][N][50].  work_dev = NULL: plain rollout of
 * the controls (alpha ignored).  chi [B][N+1][4] (node 0 is kept), u [B][N][6], alpha [B]. */
int eepacc_nlp_rollout(eepacc_nlp_handle* h, int B, const double* alpha_dev, const double* chi_dev, const double* u_dev,
                       const double* work_dev, double* chi_new_dev, double* u_new_dev, void* stream);


/* Per-route reductions of an interior-point iteration over the rows_per_route = N * R rows of each route (route-major
 * arrays r, t, lam, jdy [B][N][R]; mu, tau, alpha [B]).
 * eepacc_nlp_steprule: step dt = -(r + t) - jdy and dlam = (mu/t + (lam/t)(r + t) + (lam/t) jdy) - lam (written if jdy is
 * given) and out [B][8] = fraction-to-the-boundary step lengths a_p, a_d (<= 1), sum of residuals r + t of rows that do
 * not hold (r + t > 1e-9 (1 + t)), largest new multiplier on those rows, sum log t, largest such residual, max lam t,
 * max |lam t - mu|  (without jdy: slot 0 = min lam t, slots 4..7 = the convergence measures of the current point).
 * eepacc_nlp_trial: slacks of a trial point (rows that hold: t = -r_trial; the others: max(-r_trial, t + alpha dt)) and
 * out [B][3] = all slacks keep the fraction-to-the-boundary distance (1 / 0), sum of residuals of rows that do not
 * hold, sum log t_trial. */
int eepacc_nlp_steprule(int device, int B, int rows_per_route, const double* r_dev, const double* t_dev, const double* lam_dev,
                        const double* jdy_dev, const double* mu_dev, const double* tau_dev, double* dt_dev, double* dlam_dev,
                        double* out_dev, void* stream);
int eepacc_nlp_trial(int device, int B, int rows_per_route, const double* r_dev, const double* t_dev, const double* dt_dev,
                     const double* r_trial_dev, const double* alpha_dev, const double* tau_dev, double* t_trial_dev,
                     double* out_dev, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * The solver: `sol = solver('x0',z0,'lbx',...,'ubg',...)` of ABO/RunOpt_NLP.m:505-510 (IPOPT with the options of :246-252)
 * for a batch of routes, entirely on the device.
 * ------------------------------------------------------------------------------------------------------------------- */

/* Options; zero / negative entries select the defaults in brackets. */
typedef struct eepacc_nlp_options {
    int32_t max_iter;        /* OPTsettings.NLPmaxIter (RunOpt_NLP.m:247)                                      [1500] */
    int32_t restarts;        /* re-centrings of a route whose line search fails at every Levenberg term; < 0:   [3]    */
    int32_t max_ls;          /* step halvings per factorisation                                                [4]    */
    int32_t phase1_iter;     /* eepacc_run_nlp_host: iteration budget of the first phase (exact tables); routes without a KKT
                                point after it continue in the second phase (kink_eps_s) with the rest of max_iter  [1500] */
    double  tol;             /* KKT tolerance: dual residual, constraint violation, complementarity            [1e-7] */
    double  mu_init;         /* first barrier parameter                                                        [1.0]  */
    double  mu_min;          /* its floor                                                                      [1e-9] */
    double  obj_scale;       /* objective scaling of the iteration (the reported objective is unscaled)        [1e-5] */
    double  margin;          /* slack margin of the start point                                                [1.0]  */
    double  kink_eps_s;      /* > 0: eepacc_nlp_solve works on a copy of the position lookups (speed limit, curvature, stop
                                profile, velocity incentive) whose kinks are rounded over +- this many metres (C^1: a parabola
                                between two extra knots); 0: the exact piecewise-linear tables.  eepacc_run_nlp_host solves
                                with the exact tables first and uses the rounded copy (0: 1e-2 m; < 0: never) only for routes
                                none of whose starts reaches a KKT point -- measured cause: the minimiser pins a node on a knot
                                (the end of the 1 m speed-limit ramp), where the piecewise-linear problem has no KKT point */
    double  kink_eps_v;      /* reserved (the ISO speed tables stay exact)                                                  */
} eepacc_nlp_options;

/* Interior-point solve of B problem instances.  Every instance is one route with one start; the
 * iteration (Newton system: eepacc_nlp_newton, factorisation: eepacc_nlp_riccati, closed-loop rollout: eepacc_nlp_rollout,
 * rows / step rules / merit) and all its per-route decisions -- barrier parameter, Levenberg term, fraction-to-the-boundary
 * step lengths, l1 merit, accept / reject, re-centring restoration, termination -- run on the device; the host only
 * launches and reads one counter every few rounds.
 *   s_tv_dev   [B][N]     lead-vehicle position per instance (route-major)
 *   group_dev  [B] / NULL instances of one group are starts of the same problem: the first that reaches a KKT point ends
 *                         its group (ids 0 .. n_groups-1)
 *   chi0_dev   [B][4]     node 0: (s_init, v_init, p_0, 0), p_0 = acceleration under zero force (RunOpt_NLP.m:337,371-375)
 *   forces_dev [B][N][2]  start: (Fm, Fb <= 0) per interval; states by rollout, slacks `margin` above what the rows need
 * Outputs: chi_dev [B][N+1][4] (s, v, p, j per node), u_dev [B][N][6] (Fm, Fb, xi_v, xi_h, xi_s, xi_f), J_dev [B]
 * (objective), status_dev [B] (0 KKT point to `tol` = IPOPT's Solve_Succeeded, 1 iteration limit, 2 no acceptable step at
 * any Levenberg term after the restoration attempts), iters_dev [B] / NULL, kkt_dev [B][6] / NULL (dual residual,
 * constraint violation, complementarity at the last convergence test; barrier parameter, Levenberg term, restorations used), *ticks_out / NULL (rounds of launches). */
int eepacc_nlp_solve(eepacc_nlp_handle* h, int B, const double* s_tv_dev, const int32_t* group_dev, int n_groups,
                     const double* chi0_dev, const double* forces_dev, const eepacc_nlp_options* options,
                     double* chi_dev, double* u_dev, double* J_dev, int32_t* status_dev, int32_t* iters_dev,
                     double* kkt_dev, int32_t* ticks_out, void* stream);

/* B1 -- `NLPsol = RunOpt_NLP(OPTsettings)` (ABO/Main.m:97, ABO/RunOpt_NLP.m:1) for n_routes routes that share the handle's
 * route tables and differ in their lead trace s_tv_host [n_routes][N] (sample k = s_tv(k+1), :488-499).  Host pointers in
 * and out (what a MEX gateway holds).  The reference starts IPOPT from z0 = 0 (:348); here every route gets n_starts
 * car-following force trajectories (look-ahead samples / response time per start: eepacc_nlp_car_following_start_host)
 * solved side by side as one group, or -- start_forces_host [n_routes][N][2] given -- one warm start.
 * Outputs per route: chi_host [n_routes][N+1][4], u_host [n_routes][N][6], J_host, status_host (as eepacc_nlp_solve),
 * iters_host / NULL, start_host / NULL (index of the winning start; n_starts + index when the route was solved in the
 * second phase with rounded table kinks, see eepacc_nlp_options.kink_eps_s), all_J_host / all_status_host [n_routes][n_starts] / NULL.
 * Winner: lowest objective among the starts at a KKT point; if none, among the primal-feasible ones; else the smallest
 * constraint violation. */
int eepacc_run_nlp_host(eepacc_nlp_handle* h, int n_routes, const double* s_tv_host, double s_init, double v_init, int n_starts,
                        const int32_t* start_lookahead, const double* start_tau, const double* start_forces_host,
                        const eepacc_nlp_options* options, double* chi_host, double* u_host, double* J_host,
                        int32_t* status_host, int32_t* iters_host, int32_t* start_host, double* all_J_host,
                        int32_t* all_status_host);
int eepacc_nlp_car_following_start_host(eepacc_nlp_handle* h, const double* s_tv_host, double s_init, double v_init, int lookahead,
                                        double tau, double* forces_host /* [N][2] */);

/* Host-side problem construction of RunOpt_NLP.m:63-184 from the route description of eepacc_settings (speed limits,
 * curvature, slope, stops, traffic lights and their constants): stop and traffic-light profiles and the velocity-incentive
 * profile through minPWA / SaturateSlopePWA / FixCrossingPWA / SimplifyPWA (ABO/Functions/PWA_function_manipulation/).
 * Fills *p (tables point into *owner, free it with eepacc_nlp_tables_free after eepacc_nlp_create).  No GPU needed.
 * W_NLP [7] (Settings.m:12-29), b [21] (b_fifthOrder, or b_quadr followed by 15 zeros). */
typedef struct eepacc_nlp_tables eepacc_nlp_tables;
int  eepacc_nlp_problem_from_settings(eepacc_nlp_tables** owner, eepacc_nlp_problem* p, const eepacc_settings* S,
                                      const double W_NLP[7], const double b[21], double Ts, double t_sim);
void eepacc_nlp_tables_free(eepacc_nlp_tables* t);

/* Derived quantities of optSol (RunOpt_NLP.m:545-605) on the host: rpm, P (fifth-order surface), E, a, Tm [N] and, if
 * cost is given, the seven running cost series [7][N] (P, a, j, xi_v, xi_h, xi_s, xi_f). */
int eepacc_nlp_postprocess_host(const eepacc_vehicle* V, const double b_fifthOrder[21], const double W_NLP[7], double Ts, int N,
                                const double* v_opt, const double* Fm_opt, const double* j_opt, const double* slacks,
                                double* rpm, double* P, double* E, double* a, double* Tm, double* cost);

#ifdef __cplusplus
}
#endif
#endif
