// Internal interface of the batched dense QP kernel (eepacc_qp_dense.hip).
#ifndef EEPACC_QP_DENSE_H
#define EEPACC_QP_DENSE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct eepacc_qp_args {
    int B, nV, nC;
    const double *H, *g, *A, *lba, *uba, *lbx, *ubx, *x0;   // device, instance-major
    double *x, *cost;
    int32_t *status, *iters;
    int* counter;          // device int, zeroed before the launch (work distribution)
    int* rho_k;            // [B] in/out hint: exponent k of the regularisation found last time (may be NULL)
    double* ws;            // grid * ws_stride doubles
    size_t ws_stride;
    double rho_rel;        // <= 0: 1e-7
    int max_prox;          // <= 0: 8
};

size_t eepacc_qp_dense_ws_doubles(int nV);
size_t eepacc_qp_dense_lds_bytes(int nV, int nC);
hipError_t eepacc_qp_dense_launch(const eepacc_qp_args& a, int grid, hipStream_t stream);

#endif
