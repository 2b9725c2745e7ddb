/*
 * RunOpt_BLMPC.c -- MEX gateway: optSol = RunOpt_BLMPC(OPTsettings)      (ABO/RunOpt_BLMPC.m:1, ABO/Main.m:97)
 * Drop-in for the baseline controller's closed loop; see eepacc_mex_common.h for the contract and the build line
 *     mex -I../include RunOpt_BLMPC.c -L../eepacc_mpc_casadi_matlab_amd -leepacc
 * The baseline controller is a handle created with bl_mode = 1, run through eepacc_run_blmpc_host (include/eepacc.h):
 * horizon BL_N_hor with the uniform step Tvec(1) (RunOpt_BLMPC.m:17,20), ego estimator BL_trajEstSett
 * (EstimateVehicleTrajectory.m:25-29), weights W_BL (CreateQP_BL.m:36-39), comfort limits BL_*_Lim*Vel
 * (EstimateRouteAndComfortBounds.m:41-46).
 */
#include <time.h>
#include "eepacc_mex_common.h"

static eepacc_handle* g_handle = NULL;
static void at_exit(void) { if (g_handle) { eepacc_destroy(g_handle); g_handle = NULL; } }

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 1 || !mxIsStruct(prhs[0]) || nlhs > 1)
        mexErrMsgIdAndTxt("eepacc:usage", "usage: optSol = RunOpt_BLMPC(OPTsettings)");
    const mxArray* O = prhs[0];
    eepacc_mex_inputs in;
    emx_read_inputs(O, 0, &in);                     /* everything the controllers share (W_AB is read but not used) */
    eepacc_settings* S = &in.S;
    int n;
    /* the baseline controller's own view of the settings */
    S->bl_mode = 1;
    S->N_hor = (int)emx_scalar(O, "BL_N_hor");                                  /* RunOpt_BLMPC.m:17 */
    if (S->N_hor < 2 || S->N_hor > EEPACC_MAX_HORIZON) mexErrMsgIdAndTxt("eepacc:badField", "BL_N_hor must be in [2, %d]", EEPACC_MAX_HORIZON);
    double* Tv = (double*)mxMalloc(sizeof(double) * (size_t)S->N_hor);
    for (int i = 0; i < S->N_hor; ++i) Tv[i] = in.Ts;                           /* Ts = Tvec(1), :20 (BL_Ts, Settings.m:138) */
    S->Tvec = Tv;
    mxFree(in.Mb);
    in.Mb = (int32_t*)mxCalloc((size_t)S->N_hor, sizeof(int32_t));              /* no move blocking in RunOpt_BLMPC */
    S->Mb = in.Mb;
    S->paramEstSetting = (int)emx_scalar(O, "BL_trajEstSett");                  /* EstimateVehicleTrajectory.m:27 */
    {
        const double* W = emx_vector(O, "W_BL", &n, 1);                         /* CreateQP_BL.m:27,36-39 */
        if (n != 4) mexErrMsgIdAndTxt("eepacc:badField", "W_BL must have 4 entries [w_v, w_a, w_j, w_f]");
        memcpy(S->W_BL, W, sizeof(double) * 4);
    }
    S->BL_a_LimLowVel = emx_scalar(O, "BL_a_LimLowVel"); S->BL_a_LimHighVel = emx_scalar(O, "BL_a_LimHighVel");
    S->BL_j_LimLowVel = emx_scalar(O, "BL_j_LimLowVel"); S->BL_j_LimHighVel = emx_scalar(O, "BL_j_LimHighVel");
    S->bl_lp_eps = 0.0; S->state_bound_tol = 0.0;                               /* library defaults */
    at_exit();
    mexAtExit(at_exit);
    if (eepacc_create(&g_handle, S, &in.V, 0, 1) != EEPACC_OK)
        mexErrMsgIdAndTxt("eepacc:create", "%s", eepacc_last_error());
    const int ns = in.n_steps;
    double* traj = (double*)mxMalloc(sizeof(double) * (size_t)ns * EEPACC_OUT_N);
    int32_t* status = (int32_t*)mxMalloc(sizeof(int32_t) * (size_t)ns);
    const clock_t c0 = clock();
    const int rc = eepacc_run_blmpc_host(g_handle, 1, ns, &in.s_init, &in.v_init, &in.a_minus1, in.s_tv, in.v_tv, traj, status);
    const double wall = (double)(clock() - c0) / CLOCKS_PER_SEC;
    at_exit();
    if (rc != EEPACC_OK) mexErrMsgIdAndTxt("eepacc:run", "%s", eepacc_last_error());
    /* optSol of RunOpt_BLMPC.m:233-234,318-345: the common builder's fields minus the slacks / DistHor / cost the
     * baseline loop does not keep */
    mxArray* sol = emx_build_optsol(&in, traj, status, wall, 0, NULL, NULL, NULL);
    static const char* drop[] = {"xi_v_opt", "xi_h_opt", "xi_s_opt", "xi_f_opt", "DistHor", "cost"};
    for (unsigned i = 0; i < sizeof drop / sizeof drop[0]; ++i) {
        const int f = mxGetFieldNumber(sol, drop[i]);
        if (f >= 0) { mxDestroyArray(mxGetFieldByNumber(sol, 0, f)); mxRemoveField(sol, f); }
    }
    plhs[0] = sol;
    mxFree(traj); mxFree(status); mxFree(in.Mb); mxFree(Tv); if (in.TLLoc) mxFree(in.TLLoc);
}
