/*
 * RunOpt_ABMPC.c -- MEX gateway: optSol = RunOpt_ABMPC(OPTsettings)      (ABO/RunOpt_ABMPC.m:1, ABO/Main.m:115)
 * Drop-in for the acceleration-based MPC closed loop; see eepacc_mex_common.h for the contract and the build line.
 */
#include <time.h>
#include "eepacc_mex_common.h"

static eepacc_handle* g_handle = NULL;
static void at_exit(void) { if (g_handle) { eepacc_destroy(g_handle); g_handle = NULL; } }

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 1 || !mxIsStruct(prhs[0]) || nlhs > 1)
        mexErrMsgIdAndTxt("eepacc:usage", "usage: optSol = RunOpt_ABMPC(OPTsettings)");
    eepacc_mex_inputs in;
    emx_read_inputs(prhs[0], 0, &in);
    /* the settings can change between calls (Main.m sweeps): one handle per call, destroyed at the end or at exit */
    at_exit();
    mexAtExit(at_exit);
    if (eepacc_create(&g_handle, &in.S, &in.V, 0, 1) != EEPACC_OK)
        mexErrMsgIdAndTxt("eepacc:create", "%s", eepacc_last_error());            /* malformed / unsupported settings */
    const int n = in.n_steps;
    double* traj = (double*)mxMalloc(sizeof(double) * (size_t)n * EEPACC_OUT_N);
    int32_t* status = (int32_t*)mxMalloc(sizeof(int32_t) * (size_t)n);
    const clock_t c0 = clock();
    const int rc = eepacc_run_abmpc_host(g_handle, 1, n, &in.s_init, &in.v_init, &in.a_minus1, in.s_tv, in.v_tv, traj, status);
    const double wall = (double)(clock() - c0) / CLOCKS_PER_SEC;
    at_exit();
    if (rc != EEPACC_OK) mexErrMsgIdAndTxt("eepacc:run", "%s", eepacc_last_error());
    /* cost series: the reference reads W(1..5) whatever the length of W_AB and uses W(5) twice (RunOpt_ABMPC.m:381-388) */
    static const char* cnames[] = {"cost_a", "cost_j", "cost_xi_v", "cost_xi_h", "cost_xi_s", "cost_xi_f"};
    static const int csrc[] = {1, 2, 3, 4, 5, 6};
    double cw[6] = {in.W[0], in.W[1], in.W[2], in.W[3], in.W[4], in.W[4]};
    plhs[0] = emx_build_optsol(&in, traj, status, wall, 6, cnames, cw, csrc);
    mxFree(traj); mxFree(status); mxFree(in.Mb); if (in.TLLoc) mxFree(in.TLLoc);
}
