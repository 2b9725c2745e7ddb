/*
 * RunOpt_FBMPC.c -- MEX gateway: optSol = RunOpt_FBMPC(OPTsettings)      (ABO/RunOpt_FBMPC.m:1, ABO/Main.m:106)
 * Drop-in for the force-based MPC closed loop; see eepacc_mex_common.h for the contract and the build line.
 */
#include <time.h>
#include "eepacc_mex_common.h"

static eepacc_handle* g_handle = NULL;
static void at_exit(void) { if (g_handle) { eepacc_destroy(g_handle); g_handle = NULL; } }

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 1 || !mxIsStruct(prhs[0]) || nlhs > 1)
        mexErrMsgIdAndTxt("eepacc:usage", "usage: optSol = RunOpt_FBMPC(OPTsettings)");
    eepacc_mex_inputs in;
    emx_read_inputs(prhs[0], 1, &in);
    at_exit();
    mexAtExit(at_exit);
    if (eepacc_create(&g_handle, &in.S, &in.V, 0, 1) != EEPACC_OK)
        mexErrMsgIdAndTxt("eepacc:create", "%s", eepacc_last_error());
    const int n = in.n_steps;
    double* traj = (double*)mxMalloc(sizeof(double) * (size_t)n * EEPACC_OUT_N);
    int32_t* status = (int32_t*)mxMalloc(sizeof(int32_t) * (size_t)n);
    const clock_t c0 = clock();
    const int rc = eepacc_run_fbmpc_host(g_handle, 1, n, &in.s_init, &in.v_init, &in.a_minus1, in.s_tv, in.v_tv, traj, status);
    const double wall = (double)(clock() - c0) / CLOCKS_PER_SEC;
    at_exit();
    if (rc != EEPACC_OK) mexErrMsgIdAndTxt("eepacc:run", "%s", eepacc_last_error());
    /* cost series with the seven FB weights (RunOpt_FBMPC.m:362-397) */
    static const char* cnames[] = {"cost_P", "cost_a", "cost_j", "cost_xi_v", "cost_xi_h", "cost_xi_s", "cost_xi_f"};
    static const int csrc[] = {0, 1, 2, 3, 4, 5, 6};
    plhs[0] = emx_build_optsol(&in, traj, status, wall, 7, cnames, in.W, csrc);
    mxFree(traj); mxFree(status); mxFree(in.Mb); if (in.TLLoc) mxFree(in.TLLoc);
}
