// eepacc_wave.h -- wave64 primitives of the one-QP-per-wavefront kernels (gfx950): lane broadcasts, DPP
// prefix scans, arg-max reductions.  Everything stays in the VALU data path (no LDS round trips).
#pragma once
#include <hip/hip_runtime.h>

namespace eepacc {
namespace wv {

#define EEPACC_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// broadcast from a wave-uniform source lane (v_readlane)
__device__ __forceinline__ double bcast(double x, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    int lo = __builtin_amdgcn_readlane(__double2loint(x), s);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), s);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bcast_i(int x, int src) {
    return __builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(src));
}

// DPP cross-lane moves.  ctrl: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143, wave_shr:1 = 0x138,
// wave_shl:1 = 0x130 (gfx9-family encodings).  dpp_zero: lanes without a valid source read 0; dpp_keep: they keep
// their value; dpp_fill: they read `fill`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_zero(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_keep(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fill(double x, double fill) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane63(double x) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}
// value of the previous / next lane (0 at the ends)
__device__ __forceinline__ double lane_prev(double x) { return dpp_zero<0x138, 0xf>(x); }
__device__ __forceinline__ double lane_next(double x) { return dpp_zero<0x130, 0xf>(x); }

// inclusive prefix sum over the 64 lanes (Hillis-Steele inside 16-lane rows, then row broadcasts)
__device__ __forceinline__ double scan_incl(double x) {
    x += dpp_zero<0x111, 0xf>(x);
    x += dpp_zero<0x112, 0xf>(x);
    x += dpp_zero<0x114, 0xf>(x);
    x += dpp_zero<0x118, 0xf>(x);
    x += dpp_zero<0x142, 0xa>(x);
    x += dpp_zero<0x143, 0xc>(x);
    return x;
}
__device__ __forceinline__ double wave_sum(double x) { return read_lane63(scan_incl(x)); }
__device__ __forceinline__ double scan_excl(double x) { return dpp_zero<0x138, 0xf>(scan_incl(x)); }
// exclusive prefix product (lane 0 gets 1)
__device__ __forceinline__ double scan_prod_excl(double x) {
    x *= dpp_fill<0x111, 0xf>(x, 1.0);
    x *= dpp_fill<0x112, 0xf>(x, 1.0);
    x *= dpp_fill<0x114, 0xf>(x, 1.0);
    x *= dpp_fill<0x118, 0xf>(x, 1.0);
    x *= dpp_fill<0x142, 0xa>(x, 1.0);
    x *= dpp_fill<0x143, 0xc>(x, 1.0);
    return dpp_fill<0x138, 0xf>(x, 1.0);
}
__device__ __forceinline__ double wave_max(double x) {
    x = fmax(x, dpp_keep<0x111, 0xf>(x));
    x = fmax(x, dpp_keep<0x112, 0xf>(x));
    x = fmax(x, dpp_keep<0x114, 0xf>(x));
    x = fmax(x, dpp_keep<0x118, 0xf>(x));
    x = fmax(x, dpp_keep<0x142, 0xa>(x));
    x = fmax(x, dpp_keep<0x143, 0xc>(x));
    return read_lane63(x);
}
// arg-max / arg-min with integer payload; ties go to the lowest lane (deterministic)
__device__ __forceinline__ void wave_argmax(double& v, int& p) {
    const double best = wave_max(v);
    const unsigned long long mask = __ballot(v == best);
    const int src = mask ? (__ffsll((long long)mask) - 1) : 0;
    p = __builtin_amdgcn_readlane(p, src);
    v = best;
}
__device__ __forceinline__ void wave_argmin(double& v, int& p) {
    double nv = -v;
    wave_argmax(nv, p);
    v = -nv;
}

}  // namespace wv
}  // namespace eepacc
