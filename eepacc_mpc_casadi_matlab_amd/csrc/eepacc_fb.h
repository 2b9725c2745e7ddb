// Internal interface of the FBMPC kernels (eepacc_fb.hip).
#ifndef EEPACC_FB_H
#define EEPACC_FB_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "eepacc_device.h"

namespace eepacc {

struct eepacc_fb_args {
    const DevCfg* cfg;
    int B, k_step;
    int b0, nb;                 // this launch builds instances b0 .. b0+nb-1 into QP slots 0 .. nb-1
    int mode;                   // 0: per-step operator (all inputs given), 1: closed loop (plant from carry)
    // mode 0: measured state and lead data, each [B].  mode 1, k_step == 0: s, v, a_prev = s0, v0,
    // a_minus1; s_tv, v_tv = row k_step of the lead traces.
    const double *s, *v, *a_prev, *t0, *s_tv, *v_tv, *a_tv_prev;
    const double* carry;        // [5][B]: s, v, Fm, Fb, lead speed of the previous step
    double *A22, *D2;           // [N][B] carried state-space entries (A(k)/D(k) index quirk)
    const double *sp_prev, *vp_prev;   // [N+1][B] predictions of the previous step (paramEstSetting 2)
    double *H, *g, *A, *lba, *uba;   // dense QP, instance-major (A column-major nC x nV)
    double* meas;               // [5][B]: s, v, DistHor, lead speed, a_opt(k)
};

struct eepacc_fb_apply_args {
    const DevCfg* cfg;
    int B;
    const double *x, *cost;
    const int32_t* qp_status;
    const double *meas, *A22, *D2;
    double *out, *s_pred, *v_pred;
    int32_t* status;
    double* carry;              // NULL for the per-step operator
};

size_t fb_build_smem_bytes(int N);
hipError_t launch_fb_build(const eepacc_fb_args& a, int N, hipStream_t stream);
hipError_t launch_fb_apply(const eepacc_fb_apply_args& a, hipStream_t stream);

}  // namespace eepacc
#endif
