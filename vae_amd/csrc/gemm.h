// gemm.h - fp32 MFMA GEMM with generic strides, used for every Linear forward/backward.
//
//   C[m][n] (=, += or atomic+=) act( sum_k A[m][k] * B[k][n] + bias[n] )
//   A[m][k] at A + m*sam + k*sak,  B[k][n] at Bm + k*sbk + n*sbn,  C[m][n] at C + m*scm + n
//
// v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain in k order): a 64x64 block tile,
// 4 waves as 2x2, each wave one 32x32 accumulator; BK = 16 staged through LDS as
// As[k][m] / Bs[k][n] so that the per-lane operand reads (lane -> m or n, k = lane>>5)
// are conflict-free ds_read_b32.  Linear layers here have K,N in 15..288 and M = batch, so
// the tile is deliberately small; ragged edges are zero-filled on load and masked on store.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { GEMM_STORE = 0, GEMM_ACCUM = 1, GEMM_ATOMIC = 2 };

struct GemmArgs {
    const float* A;
    const float* Bm;
    float* C;
    const float* bias;
    int M, N, K;
    int64_t sam, sak, sbk, sbn, scm;
    int kchunk;  // K range per blockIdx.z (split-K); == K when gridDim.z == 1
    int act;     // VAE_ACT_NONE / VAE_ACT_LEAKY
    float slope;
    int mode;
};

#define GEMM_BM 64
#define GEMM_BN 64
#define GEMM_BK 16
#define GEMM_LD 65  // +1 pad: k-major tiles, conflict-free column writes

__global__ __launch_bounds__(256) void gemm_f32_mfma_k(GemmArgs a) {
    __shared__ float As[GEMM_BK * GEMM_LD];
    __shared__ float Bs[GEMM_BK * GEMM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * GEMM_BM, n0 = blockIdx.x * GEMM_BN;
    const int k_begin = blockIdx.z * a.kchunk;
    const int k_end = min(a.K, k_begin + a.kchunk);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // thread -> (row-in-tile, k) staging maps; pick the one whose fast index is memory-contiguous
    const bool a_kfast = (a.sak == 1);
    const bool b_kfast = (a.sbk == 1);

    for (int k0 = k_begin; k0 < k_end; k0 += GEMM_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int mm, kk;
            if (a_kfast) { kk = tid & 15; mm = (tid >> 4) + 16 * i; }
            else { mm = tid & 63; kk = (tid >> 6) + 4 * i; }
            const int gm = m0 + mm, gk = k0 + kk;
            As[kk * GEMM_LD + mm] = (gm < a.M && gk < k_end) ? a.A[gm * a.sam + gk * a.sak] : 0.f;
            int nn, kb;
            if (b_kfast) { kb = tid & 15; nn = (tid >> 4) + 16 * i; }
            else { nn = tid & 63; kb = (tid >> 6) + 4 * i; }
            const int gn = n0 + nn, gkb = k0 + kb;
            Bs[kb * GEMM_LD + nn] = (gn < a.N && gkb < k_end) ? a.Bm[gkb * a.sbk + gn * a.sbn] : 0.f;
        }
        __syncthreads();
        const int kh = lane >> 5, r = lane & 31;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK; kk += 2) {
            const float av = As[(kk + kh) * GEMM_LD + wm * 32 + r];
            const float bv = Bs[(kk + kh) * GEMM_LD + wn * 32 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= a.N) return;
    const float bv = (a.bias != nullptr && blockIdx.z == 0) ? a.bias[n] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (m >= a.M) continue;
        float v = acc[reg] + bv;
        if (a.act == VAE_ACT_LEAKY) v = v > 0.f ? v : v * a.slope;
        float* cp = a.C + m * a.scm + n;
        if (a.mode == GEMM_STORE) *cp = v;
        else if (a.mode == GEMM_ACCUM) *cp += v;
        else atomicAdd(cp, v);
    }
}
