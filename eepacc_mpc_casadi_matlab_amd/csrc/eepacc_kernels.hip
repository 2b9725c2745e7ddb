// eepacc_kernels.hip -- hand-written HIP (gfx950) kernels of the batched ABMPC step.
//
// One QP instance per 64-lane wavefront: horizon stage k lives on lane k (terminal stage on
// lane N), the small inverse-KKT block of the working set and all per-stage vectors are staged
// in LDS.  The horizon condensing is never materialised: every constraint row of the reference
// QP (ABO/Functions/MPCs/CreateQP_AB.m:256-387) is  al*s_k + be*v_k + ga*a_k + de*a_{k-1} - xi <= b
// and the condensed double integrator (ABO/RunOpt_ABMPC.m:74-82 +
// ABO/Functions/MPCs/TransformToDenseFormulation.m:46-68) turns into wave-level prefix /
// suffix scans.  The slack columns are eliminated exactly (capped-multiplier groups, compliant
// row for the quadratic slack); the dense active-set solve is a dual (Goldfarb-Idnani type)
// method on the N x N acceleration block.  DESIGN.md has the derivation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "eepacc_device.h"
#include "eepacc_stage.h"
#include "../../include/eepacc.h"

#define EEPACC_IMPL_NS nomb
#define EEPACC_IMPL_MB false
#define EEPACC_IMPL_BL false
#include "eepacc_ab_impl.inc"
#undef EEPACC_IMPL_NS
#undef EEPACC_IMPL_MB
#define EEPACC_IMPL_NS withmb
#define EEPACC_IMPL_MB true
#include "eepacc_ab_impl.inc"
#undef EEPACC_IMPL_NS
#undef EEPACC_IMPL_MB
#undef EEPACC_IMPL_BL
// baseline controller (RunOpt_BLMPC): the same kernels with CreateQP_BL's row grouping
#define EEPACC_IMPL_NS blc
#define EEPACC_IMPL_MB false
#define EEPACC_IMPL_BL true
#include "eepacc_ab_impl.inc"
#undef EEPACC_IMPL_NS
#undef EEPACC_IMPL_MB
#undef EEPACC_IMPL_BL

// ICE-map fuel term (CreateQP_AB.m:154-159): step-varying Hessian, built and inverted in LDS every step
#define EEPACC_IMPL_NS ice
#define EEPACC_IMPL_MB false
#define EEPACC_IMPL_BL false
#define EEPACC_IMPL_ICE true
#include "eepacc_ab_impl.inc"
#undef EEPACC_IMPL_NS
#undef EEPACC_IMPL_MB
#undef EEPACC_IMPL_BL
#undef EEPACC_IMPL_ICE

// ----------------------------------------------------------------------------------------------
// host-side launchers used by eepacc_capi.cpp
namespace eepacc {

// working-set capacity: rigid rows are linearly independent, so m <= N (+ terminal rows)
#ifndef EEPACC_MMAX_SMALL
#define EEPACC_MMAX_SMALL 34
#endif
constexpr int kMMaxSmall = EEPACC_MMAX_SMALL, kNSSmall = 32;     // N <= 32: 8 waves / CU (4 per block, 2 blocks)
constexpr int kBlocksSmall = kMMaxSmall <= 32 ? 3 : 2;
// N <= 63: 44.6 KB of LDS per wave (He packed 16.6 KB, P 17.7 KB), 3 waves per CU (round 2: a full He of 32 KB allowed 2).
// Trading working-set capacity for more does not work: with a capacity of 50 rigid rows (-DEEPACC_MMAX_LARGE=50) the S2
// workload at N = 60 overflows the working set on 15 % of the steps (measured), so the full N + 2 stays.
#ifndef EEPACC_MMAX_LARGE
#define EEPACC_MMAX_LARGE 66
#endif
#ifndef EEPACC_WPB_LARGE
#define EEPACC_WPB_LARGE 3
#endif
constexpr int kMMaxLarge = EEPACC_MMAX_LARGE, kNSLarge = 64, kWpbLarge = EEPACC_WPB_LARGE;
constexpr int kChunkStepsDefault = nomb::kChunkStepsDefault;

#ifdef EEPACC_AB_TIMING
extern "C" int eepacc_debug_ab_prof(unsigned long long* out, int reset) {
    unsigned long long z[24] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nomb::g_ab_prof), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(nomb::g_ab_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

size_t ab_smem_bytes(int N) {
    return N <= kNSSmall ? nomb::wave_bytes(sizeof(nomb::WaveMem<kMMaxSmall, kNSSmall>), kNSSmall) * 4
                         : nomb::wave_bytes(sizeof(nomb::WaveMem<kMMaxLarge, kNSLarge>), kNSLarge) * kWpbLarge;
}

// the kernel of the namespace with / without move blocking, small or large horizon
#define EEPACC_LAUNCH_NS(NSP, KERNEL, MM, NSV, WPB, GRID, ...)                                                \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(NSP::KERNEL<MM, NSV, WPB>), dim3(GRID), dim3(64 * WPB), ab_smem_bytes(N), stream, __VA_ARGS__)
#define EEPACC_LAUNCH(KERNEL, MM, NSV, WPB, GRID, ...)                                                        \
    do { if (variant == 3) EEPACC_LAUNCH_NS(ice, KERNEL, MM, NSV, WPB, GRID, __VA_ARGS__);                    \
         else if (variant == 2) EEPACC_LAUNCH_NS(blc, KERNEL, MM, NSV, WPB, GRID, __VA_ARGS__);               \
         else if (variant == 1) EEPACC_LAUNCH_NS(withmb, KERNEL, MM, NSV, WPB, GRID, __VA_ARGS__);            \
         else EEPACC_LAUNCH_NS(nomb, KERNEL, MM, NSV, WPB, GRID, __VA_ARGS__); } while (0)

// MPC steps per work unit of the closed-loop kernels: 16 (EEPACC_CHUNK overrides), fewer when the launch is so short
// that the resident waves would otherwise get fewer than about eight units each (tail imbalance)
int pick_chunk_steps(int n_steps, int B, int resident_waves) {
    static int forced = -1;
    if (forced < 0) {
        const char* ev = getenv("EEPACC_CHUNK");
        forced = (ev && atoi(ev) > 0) ? atoi(ev) : 0;
    }
    if (forced > 0) return forced;
    long long per = ((long long)n_steps * B) / (8LL * (resident_waves > 0 ? resident_waves : 1));
    int c = per > kChunkStepsDefault ? kChunkStepsDefault : (int)per;
    return c < 2 ? 2 : c;
}

// per-wave scratch of the ICE variant (base inverse of the step): one NS x NS block for every wave a launch can have
size_t ab_hb_doubles(int N, int B, int num_cus) {
    if (N > kNSSmall) return 0;
    const size_t step_waves = (size_t)((B + 3) / 4) * 4, run_waves = (size_t)num_cus * kBlocksSmall * 4;
    return (step_waves > run_waves ? step_waves : run_waves) * kNSSmall * kNSSmall;
}

hipError_t launch_ab_step(const DevCfg* dC, int N, int variant, int B, const double* s, const double* v, const double* a_prev,
                          const double* t0, const double* s_tv, const double* v_tv, const double* a_tv_prev,
                          unsigned long long* codes, double* out, double* s_pred, double* v_pred,
                          int32_t* status, int32_t* iters, hipStream_t stream) {
    if (N > kNSSmall) EEPACC_LAUNCH(k_ab_step, kMMaxLarge, kNSLarge, kWpbLarge, (B + kWpbLarge - 1) / kWpbLarge, dC, B, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, codes, out, s_pred, v_pred, status, iters);
    else EEPACC_LAUNCH(k_ab_step, kMMaxSmall, kNSSmall, 4, (B + 3) / 4, dC, B, s, v, a_prev, t0, s_tv, v_tv, a_tv_prev, codes, out, s_pred, v_pred, status, iters);
    return hipGetLastError();
}

hipError_t launch_run_abmpc(const DevCfg* dC, int N, int variant, int B, int k_start, int n_steps, const double* s0,
                            const double* v0, const double* a_m1, const double* s_tv, const double* v_tv,
                            double* carry, unsigned long long* codes, double* traj,
                            int32_t* status, int32_t* iters_total, int* work_counter, int* done, int* err_word, int num_cus,
                            hipStream_t stream) {
    hipError_t e = hipMemsetAsync(work_counter, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(done, 0, sizeof(int) * (size_t)B, stream);
    if (e != hipSuccess) return e;
    if (iters_total) {
        e = hipMemsetAsync(iters_total, 0, sizeof(int32_t) * (size_t)B, stream);
        if (e != hipSuccess) return e;
    }
    const int kChunkSteps = pick_chunk_steps(n_steps, B, num_cus * (N > kNSSmall ? kWpbLarge : 4 * kBlocksSmall));
    // bound of the inter-unit spin wait (a debug hook lowers it to exercise the failure path)
    int spin_limit = 1 << 26;
    if (const char* ev = getenv("EEPACC_DEBUG_SPIN_LIMIT")) spin_limit = atoi(ev);
    const int n_units = ((n_steps + kChunkSteps - 1) / kChunkSteps) * B;
    // one chip-filling wave of blocks: LDS admits 8 (small) / 2 (large) waves per CU
    if (N > kNSSmall) {
        int grid = num_cus * 1, need = (n_units + kWpbLarge - 1) / kWpbLarge;
        if (grid > need) grid = need;
        EEPACC_LAUNCH(k_run_abmpc, kMMaxLarge, kNSLarge, kWpbLarge, grid, dC, B, k_start, n_steps, s0, v0, a_m1, s_tv, v_tv, carry, codes, traj, status, iters_total, work_counter, done, kChunkSteps, err_word, spin_limit);
    } else {
        int grid = num_cus * kBlocksSmall, need = (n_units + 3) / 4;
        if (grid > need) grid = need;
        EEPACC_LAUNCH(k_run_abmpc, kMMaxSmall, kNSSmall, 4, grid, dC, B, k_start, n_steps, s0, v0, a_m1, s_tv, v_tv, carry, codes, traj, status, iters_total, work_counter, done, kChunkSteps, err_word, spin_limit);
    }
    return hipGetLastError();
}

hipError_t launch_postprocess(const DevCfg* dC, int B, int n_steps, const double* traj, double* rpm, double* Tm,
                              double* P, double* E, hipStream_t stream) {
    hipLaunchKernelGGL(nomb::k_postprocess, dim3((B + 127) / 128), dim3(128), 0, stream, dC, B, n_steps, traj, rpm, Tm, P, E);
    return hipGetLastError();
}

hipError_t set_max_smem() {
    const void* fns[16] = {reinterpret_cast<const void*>(&nomb::k_ab_step<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&nomb::k_ab_step<kMMaxLarge, kNSLarge, kWpbLarge>),
                          reinterpret_cast<const void*>(&nomb::k_run_abmpc<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&nomb::k_run_abmpc<kMMaxLarge, kNSLarge, kWpbLarge>),
                          reinterpret_cast<const void*>(&withmb::k_ab_step<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&withmb::k_ab_step<kMMaxLarge, kNSLarge, kWpbLarge>),
                          reinterpret_cast<const void*>(&withmb::k_run_abmpc<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&withmb::k_run_abmpc<kMMaxLarge, kNSLarge, kWpbLarge>),
                          reinterpret_cast<const void*>(&blc::k_ab_step<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&blc::k_ab_step<kMMaxLarge, kNSLarge, kWpbLarge>),
                          reinterpret_cast<const void*>(&blc::k_run_abmpc<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&blc::k_run_abmpc<kMMaxLarge, kNSLarge, kWpbLarge>),
                          reinterpret_cast<const void*>(&ice::k_ab_step<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&ice::k_ab_step<kMMaxLarge, kNSLarge, kWpbLarge>),
                          reinterpret_cast<const void*>(&ice::k_run_abmpc<kMMaxSmall, kNSSmall, 4>),
                          reinterpret_cast<const void*>(&ice::k_run_abmpc<kMMaxLarge, kNSLarge, kWpbLarge>)};
    for (int i = 0; i < 16; ++i) {
        // 160 KB of LDS per CU minus the kernel's static index table (one ushort per packed entry of P)
        const int mm = (i & 1) ? kMMaxLarge : kMMaxSmall;
        const int dyn = 160 * 1024 - ((mm * (mm + 1) / 2 * 2 + 255) & ~255);
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace eepacc
