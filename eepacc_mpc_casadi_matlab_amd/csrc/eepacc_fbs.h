// Internal interface of the structured FBMPC kernels (eepacc_fbs.hip).
#ifndef EEPACC_FBS_H
#define EEPACC_FBS_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "eepacc_device.h"

namespace eepacc {

// per-instance state the structured FBMPC path carries between launches (HBM, instance-major)
constexpr int kFbsStateDoubles = 64 + 64 + 64 + 64 + 64;   // codes | A22 | D2 | s_pred | v_pred

struct fbs_step_args {          // B2: one step (ABO/RunOpt_FBMPC.m:204-320)
    const DevCfg* cfg;
    int B, k_step;
    const double *s, *v, *a_prev, *t0, *s_tv, *v_tv, *a_tv_prev;   // [B]
    double* state;              // [B][kFbsStateDoubles]
    double* hb;                 // [waves of the launch][NS*NS] scratch: base inverse Hessian of the step
    double *out, *s_pred, *v_pred;
    int32_t *status, *iters;
};

struct fbs_run_args {           // B1: closed loop (ABO/RunOpt_FBMPC.m:161-331)
    const DevCfg* cfg;
    int B, k_start, n_steps;
    const double *s0, *v0, *a_m1, *s_tv, *v_tv;
    double* carry;              // [6][B]: s, v, Fm, Fb of the previous step, previous lead speed, t_0
    double* state;              // [B][kFbsStateDoubles]
    double* hb;                 // [waves of the launch][NS*NS] scratch
    double* traj;
    int32_t *status, *iters_total;
    int *work_counter, *done, *err_word;
    int chunk_steps, spin_limit;
    int cold;                   // debug: 1 = no working-set warm start between MPC steps
};

bool fbs_supported(const DevCfg& C);
size_t fbs_smem_bytes(int N);
size_t fbs_hb_doubles(int N, int B, int num_cus);   // scratch a launch for B instances needs
hipError_t fbs_set_max_smem();
hipError_t launch_fbs_step(const fbs_step_args& a, int N, hipStream_t stream);
hipError_t launch_fbs_run(const fbs_run_args& a, int N, int num_cus, hipStream_t stream);

}  // namespace eepacc
#endif
