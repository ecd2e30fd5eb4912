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
// ---- per-launch timing (vae_time_step) -----------------------------------------------------
// When a timer is active on this thread, every VAE_LAUNCH_CHECK records a HIP event behind the launch it checks, ON
// THE LAUNCH STREAM, together with the kernel's name and the algorithmic bytes (or flops) the launch site declared
// with VAE_TNAME / VAE_TBYTES / VAE_TFLOPS just before.  Inactive (the normal case): one thread-local pointer test.
#include <stdio.h>
#define VAE_TIMER_NAME 64
struct VaeTimer {
    hipStream_t s;
    int n, cap;
    bool failed;
    hipEvent_t* ev;  // cap + 1 events; ev[0] is recorded by the owner before the first launch
    char (*names)[VAE_TIMER_NAME];
    double* bytes;
    double* flops;
};
static thread_local VaeTimer* vae_tl_timer = nullptr;
static thread_local double vae_tl_bytes = 0.0, vae_tl_flops = 0.0;
static thread_local char vae_tl_name[VAE_TIMER_NAME] = "";
static inline void vae_timer_mark(const char* name) {
    VaeTimer* t = vae_tl_timer;
    if (t == nullptr) return;
    if (t->n < t->cap) {
        if (hipEventRecord(t->ev[t->n + 1], t->s) != hipSuccess) t->failed = true;
        snprintf(t->names[t->n], VAE_TIMER_NAME, "%s", vae_tl_name[0] ? vae_tl_name : name);
        t->bytes[t->n] = vae_tl_bytes;
        t->flops[t->n] = vae_tl_flops;
        ++t->n;
    } else {
        t->failed = true;
    }
    vae_tl_bytes = vae_tl_flops = 0.0;
    vae_tl_name[0] = 0;
}
#define VAE_TNAME(...) do { if (vae_tl_timer != nullptr) snprintf(vae_tl_name, VAE_TIMER_NAME, __VA_ARGS__); } while (0)
#define VAE_TBYTES(expr) do { if (vae_tl_timer != nullptr) vae_tl_bytes = (double)(expr); } while (0)
#define VAE_TFLOPS(expr) do { if (vae_tl_timer != nullptr) vae_tl_flops = (double)(expr); } while (0)
#define VAE_LAUNCH_CHECK(name)                                                                   \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) VAE_FAIL(VAE_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
        vae_timer_mark(name);                                                                    \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers ----------------------------------------------------------------------
// Wave-wide sum on the DPP path (row shifts inside the 16-lane rows, then the two row broadcasts): six vector
// instructions, where a __shfl_down tree is six dependent ds_bpermute round trips through the LDS crossbar
// (~1 us for the handful of sums an epilogue takes).  Fixed order => deterministic.  Valid in EVERY lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1  (lanes without a source add 0)
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of every row holds the row's sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of
// the wave (s_waitcnt vmcnt(0)): behind a chunk's copy-out that is 1-3 us of store latency per chunk which nothing
// needs: a value loaded from global memory is waited for where it is used, and stores need no wait at all.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
