// common.h - shared device helpers for libvae_amd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vae_amd.h"

#define VAE_WAVE 64

// ---- host-side error plumbing ------------------------------------------------------------
void vae_set_error(const char* fmt, ...);
#define VAE_FAIL(code, ...)         \
    do {                            \
        vae_set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)
#define VAE_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) VAE_FAIL(VAE_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define VAE_LAUNCH_CHECK(name)                                                                   \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) VAE_FAIL(VAE_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers ----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, VAE_WAVE);
    return v;  // valid in lane 0
}

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, VAE_WAVE);
    return v;  // valid in every lane
}

// Sum over a 256-thread block; result valid in thread 0.  `sm` >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();  // protect sm reuse between consecutive calls
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == VAE_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (act == VAE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// d(act)/d(pre-activation) expressed through the OUTPUT y (what the forward saved).
// LeakyReLU: y > 0 <=> pre > 0 for slope >= 0; at 0 torch uses the negative-side slope.
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    if (act == VAE_ACT_LEAKY) return y > 0.f ? 1.f : slope;
    if (act == VAE_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}
