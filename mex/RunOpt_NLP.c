/*
 * RunOpt_NLP.c -- MEX gateway: optSol = RunOpt_NLP(OPTsettings)      (ABO/RunOpt_NLP.m:1, ABO/Main.m:97)
 *
 * Drop-in for the full-route nonlinear optimisation: the problem construction of RunOpt_NLP.m:63-184 (lookup tables), the
 * solve of :505-510 (IPOPT there; the batched structured interior-point solver of libeepacc here, include/eepacc_nlp.h) and
 * the output struct of :512-605.  Reads the OPTsettings fields RunOpt_NLP.m:17-49 reads plus NLPmaxIter (:248), returns
 * the same field names and shapes; exitMessage carries IPOPT's names for the three outcomes (Solve_Succeeded,
 * Maximum_Iterations_Exceeded, Restoration_Failed).  Malformed input throws (mexErrMsgIdAndTxt); a solve that does not
 * converge does not -- like the reference it returns its last iterate and says so in exitMessage.
 *
 * Build (on a machine with MATLAB and ROCm):
 *   mex -R2017b mex/RunOpt_NLP.c -Iinclude -Leepacc_mpc_casadi_matlab_amd -leepacc
 * Only multiple shooting with the fifth-order / quadratic power fits is built (shootingMethod 1, discretizationMethod 0:
 * what Settings.m selects); other values throw.
 */
#include <time.h>
#define EEPACC_MEX_NO_CLOSED_LOOP
#include "eepacc_mex_common.h"
#include "eepacc_nlp.h"

/* casadi.interpolant('LUT','linear',...) for theta_opt: linear interpolation, linear extrapolation of the end segments */
static double lut_linear(const double* xs, const double* ys, int n, double x) {
    int i = 0, q;
    if (n < 2) return n == 1 ? ys[0] : 0.0;
    for (q = 1; q < n - 1; ++q) if (xs[q] <= x) i = q;
    {
        const double dx = xs[i + 1] - xs[i];
        const double slope = dx > 0.0 ? (ys[i + 1] - ys[i]) / dx : 0.0;
        return ys[i] + slope * (x - xs[i]);
    }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const mxArray* O;
    eepacc_settings S;
    eepacc_vehicle V;
    eepacc_nlp_problem P;
    eepacc_nlp_tables* owner = NULL;
    eepacc_nlp_handle* h = NULL;
    eepacc_nlp_options opt;
    int n, n2, N, k, i, rc;
    double *TL = NULL, b21[21], Ts, t_sim, s_init, v_init;
    const double *W, *b, *s_tv, *Tvec;
    /* the start set of the multi-start (look-ahead samples, response time; profiles/r02_nlp_start_survey.json) */
    static const int32_t la[8] = {90, 120, 200, 120, 300, 160, 120, 60};
    static const double tc[8] = {4.0, 2.0, 2.0, 4.0, 2.0, 8.0, 8.0, 2.0};
    if (nrhs != 1 || !mxIsStruct(prhs[0]) || nlhs > 1)
        mexErrMsgIdAndTxt("eepacc:usage", "usage: optSol = RunOpt_NLP(OPTsettings)");
    O = prhs[0];
    memset(&S, 0, sizeof S);
    if ((int)emx_scalar_opt(O, "shootingMethod", 1.0) != 1 || (int)emx_scalar_opt(O, "discretizationMethod", 0.0) != 0)
        mexErrMsgIdAndTxt("eepacc:notBuilt", "RunOpt_NLP: only multiple shooting with the RK4 integrator is built (shootingMethod 1, discretizationMethod 0)");
    W = emx_vector(O, "W_NLP", &n, 1);                                           /* RunOpt_NLP.m:17 */
    if (n != 7) mexErrMsgIdAndTxt("eepacc:badField", "W_NLP must have 7 entries");
    b = emx_vector(O, "b_fifthOrder", &n, 1);
    if (n != 21) mexErrMsgIdAndTxt("eepacc:badField", "b_fifthOrder must have 21 entries");
    memcpy(b21, b, sizeof b21);
    memcpy(S.b_fifthOrder, b, sizeof b21);
    if (emx_scalar_opt(O, "useFifthOrderFit_NLP", 1.0) == 0.0) {                  /* :226-236: the quadratic fit, zero-padded */
        const double* bq = emx_vector(O, "b_quadr", &n, 1);
        if (n != 6) mexErrMsgIdAndTxt("eepacc:badField", "b_quadr must have 6 entries");
        memset(b21, 0, sizeof b21);
        memcpy(b21, bq, sizeof(double) * 6);
    }
    Tvec = emx_vector(O, "Tvec", &n, 1);
    if (n < 1) mexErrMsgIdAndTxt("eepacc:badField", "Tvec is empty");
    Ts = Tvec[0];                                                                 /* :50 */
    t_sim = emx_scalar(O, "t_sim");
    s_init = emx_scalar(O, "s_init"); v_init = emx_scalar(O, "v_init");
    S.s_goal = emx_scalar(O, "s_goal"); S.h_min = emx_scalar(O, "h_min"); S.tau_min = emx_scalar(O, "tau_min");
    S.alpha_TTL = emx_scalar(O, "alpha_TTL");
    S.s_speedLim = emx_vector(O, "s_speedLim", &n, 1); S.v_speedLim = emx_vector(O, "v_speedLim", &n2, 1);
    if (n != n2) mexErrMsgIdAndTxt("eepacc:badField", "s_speedLim and v_speedLim differ in length");
    S.n_speedLim = n;
    S.s_curv = emx_vector(O, "s_curv", &n, 1); S.curvature = emx_vector(O, "curvature", &n2, 1);
    if (n != n2) mexErrMsgIdAndTxt("eepacc:badField", "s_curv and curvature differ in length");
    S.n_curv = n;
    S.s_slope = emx_vector(O, "s_slope", &n, 1); S.slope = emx_vector(O, "slope", &n2, 1);
    if (n != n2) mexErrMsgIdAndTxt("eepacc:badField", "s_slope and slope differ in length");
    S.n_slope = n;
    S.stopLoc = emx_vector(O, "stopLoc", &n, 0); S.n_stop = n;
    {   /* TLLoc: n_TL x 4 [location, phase, red, green], column-major in MATLAB -> row-major */
        const mxArray* f = mxGetField(O, 0, "TLLoc");
        if (f && !mxIsEmpty(f)) {
            const int r = (int)mxGetM(f);
            const double* p = mxGetPr(f);
            int j;
            if (mxGetN(f) != 4) mexErrMsgIdAndTxt("eepacc:badField", "TLLoc must be n x 4");
            TL = (double*)mxMalloc(sizeof(double) * 4 * (size_t)r);
            for (i = 0; i < r; ++i) for (j = 0; j < 4; ++j) TL[4 * i + j] = p[(size_t)j * r + i];
            S.TLLoc = TL; S.n_TL = r;
        }
    }
    S.stopRefDist = emx_scalar(O, "stopRefDist"); S.stopRefVelSlope = emx_scalar(O, "stopRefVelSlope");
    S.stopVel = emx_scalar(O, "stopVel"); S.TLstopVel = emx_scalar(O, "TLstopVel");
    emx_vehicle(&V);                                                              /* :52 */
    if (eepacc_nlp_problem_from_settings(&owner, &P, &S, W, b21, Ts, t_sim) != EEPACC_OK)
        mexErrMsgIdAndTxt("eepacc:tables", "%s", eepacc_last_error());
    N = P.N;
    s_tv = emx_vector(O, "s_tv", &n, 1);                                          /* :45; interval k reads s_tv(k+1), :488-499 */
    if (n < N) { eepacc_nlp_tables_free(owner); mexErrMsgIdAndTxt("eepacc:badField", "s_tv must hold t_sim/Ts = %d samples", N); }
    rc = eepacc_nlp_create(&h, &P, &V, 0);
    eepacc_nlp_tables_free(owner);
    if (rc != EEPACC_OK) mexErrMsgIdAndTxt("eepacc:create", "%s", eepacc_last_error());
    memset(&opt, 0, sizeof opt);
    opt.max_iter = (int32_t)emx_scalar_opt(O, "NLPmaxIter", 5000.0);              /* :248, Settings.m:93 */
    {
        mxArray* sol = mxCreateStructMatrix(1, 1, 0, NULL);
        double* chi = (double*)mxMalloc(sizeof(double) * (size_t)(N + 1) * 4);
        double* u = (double*)mxMalloc(sizeof(double) * (size_t)N * 6);
        double* slk = (double*)mxMalloc(sizeof(double) * (size_t)N * 4);
        double* cost = (double*)mxMalloc(sizeof(double) * (size_t)N * 7);
        double J = 0.0, wall;
        int32_t status = 1, iters = 0, start = 0;
        mxArray *ms = emx_col(N + 1), *mv = emx_col(N + 1), *mth = emx_col(N + 1), *mj = emx_col(N + 1);
        mxArray* mu[6];
        mxArray *mrpm = emx_col(N), *mP = emx_col(N), *mE = emx_col(N), *ma = emx_col(N), *mTm = emx_col(N);
        static const char* unames[6] = {"Fm_opt", "Fb_opt", "xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt"};
        static const char* cnames[7] = {"cost_P", "cost_a", "cost_j", "cost_xi_v", "cost_xi_h", "cost_xi_s", "cost_xi_f"};
        static const char* exits[3] = {"Solve_Succeeded", "Maximum_Iterations_Exceeded", "Restoration_Failed"};
        const clock_t c0 = clock();
        rc = eepacc_run_nlp_host(h, 1, s_tv, s_init, v_init, 8, la, tc, NULL, &opt, chi, u, &J, &status, &iters, &start, NULL, NULL);
        wall = (double)(clock() - c0) / CLOCKS_PER_SEC;
        eepacc_nlp_destroy(h);
        if (rc != EEPACC_OK) mexErrMsgIdAndTxt("eepacc:run", "%s", eepacc_last_error());
        for (k = 0; k <= N; ++k) {
            mxGetPr(ms)[k] = chi[4 * k]; mxGetPr(mv)[k] = chi[4 * k + 1]; mxGetPr(mj)[k] = chi[4 * k + 3];
            /* theta_{k+1} = slopeLookup(s_{k+1}), or 0 on a flat route (:360-365); theta_0 as x_init holds it (:210) */
            mxGetPr(mth)[k] = P.flat ? 0.0 : lut_linear(S.s_slope, S.slope, S.n_slope, chi[4 * k]);
        }
        for (i = 0; i < 6; ++i) {
            mu[i] = emx_col(N);
            for (k = 0; k < N; ++k) mxGetPr(mu[i])[k] = u[6 * k + i];
        }
        for (k = 0; k < N; ++k) for (i = 0; i < 4; ++i) slk[4 * k + i] = u[6 * k + 2 + i];
        if (eepacc_nlp_postprocess_host(&V, S.b_fifthOrder, W, Ts, N, mxGetPr(mv), mxGetPr(mu[0]), mxGetPr(mj), slk, mxGetPr(mrpm),
                                        mxGetPr(mP), mxGetPr(mE), mxGetPr(ma), mxGetPr(mTm), cost) != EEPACC_OK)
            mexErrMsgIdAndTxt("eepacc:post", "%s", eepacc_last_error());
        /* the velocity-incentive profile saved for plotting (:181-182) */
        {
            mxArray *a1 = mxCreateDoubleMatrix(1, (mwSize)P.n_vinc, mxREAL), *a2 = mxCreateDoubleMatrix(1, (mwSize)P.n_vinc, mxREAL);
            /* the tables were freed with `owner`; rebuild the two arrays the struct carries */
            eepacc_nlp_tables* o2 = NULL; eepacc_nlp_problem P2;
            if (eepacc_nlp_problem_from_settings(&o2, &P2, &S, W, b21, Ts, t_sim) == EEPACC_OK) {
                for (k = 0; k < P2.n_vinc; ++k) { mxGetPr(a1)[k] = P2.s_vinc[k]; mxGetPr(a2)[k] = P2.v_vinc[k]; }
                eepacc_nlp_tables_free(o2);
            }
            emx_set(sol, "s_velInc", a1); emx_set(sol, "v_velInc", a2);
        }
        emx_set(sol, "tSolve", mxCreateDoubleScalar(wall));                       /* :509 */
        emx_set(sol, "exitMessage", mxCreateString(exits[status < 0 || status > 2 ? 1 : status]));   /* :510 */
        emx_set(sol, "s_opt", ms); emx_set(sol, "v_opt", mv); emx_set(sol, "theta_opt", mth); emx_set(sol, "j_opt", mj);
        for (i = 0; i < 6; ++i) emx_set(sol, unames[i], mu[i]);
        emx_set(sol, "P_opt", mP); emx_set(sol, "E_opt", mE); emx_set(sol, "a_opt", ma); emx_set(sol, "Tm_opt", mTm);
        emx_set(sol, "rpm_opt", mrpm);
        for (i = 0; i < 7; ++i) {                                                 /* row vectors, as the reference's loop grows them (:593-601) */
            mxArray* c = mxCreateDoubleMatrix(1, (mwSize)N, mxREAL);
            for (k = 0; k < N; ++k) mxGetPr(c)[k] = cost[(size_t)i * N + k];
            emx_set(sol, cnames[i], c);
        }
        /* extras the reference does not return: objective, iterations, the winning start of the multi-start */
        emx_set(sol, "J", mxCreateDoubleScalar(J));
        emx_set(sol, "iterations", mxCreateDoubleScalar((double)iters));
        emx_set(sol, "start_index", mxCreateDoubleScalar((double)start));
        plhs[0] = sol;
        mxFree(chi); mxFree(u); mxFree(slk); mxFree(cost);
    }
    if (TL) mxFree(TL);
}
