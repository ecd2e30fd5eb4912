// wide.h - convolution layers too wide for the LDS-resident kernels of igemm.h (their [k][n] weight image or
// their accumulator tiles do not fit): classic implicit GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32) with a K loop.
//
// A workgroup (256 threads, 4 waves) owns a 64 x 64 output tile; every K step stages 16 k-columns of both
// operands in LDS ([row][17] layout: conflict-free for the global->LDS fill and for the MFMA operand reads),
// double-buffered, the next step's global loads in flight during the MFMAs of the current one.
//
//   wide_gather_k<TR> : y[b,n,oh,ow] = act(bias[n] + sum_k x(gathered) * w)   (= conv_gather_k / convT_gather_k
//                       of conv.h: Conv forward + ConvT input gradient; TR: ConvT forward + Conv input gradient).
//                       GEMM: D[channel][pixel], K = (ci, tap); transposed gathers run per output-parity class.
//   wide_wgrad_k      : dw[a][c][kh][kw] += sum_{b,oh,ow} S[b,a,oh,ow] * G[b,c,oh*sh-ph+kh,ow*sw-pw+kw]
//                       (= conv_wgrad_k).  GEMM: D[a][(c,tap)], K = pixels, split over grid.x; every slice writes its
//                       own copy of dw and wide_wgrad_reduce_k adds them in a fixed order (a caller that passes no
//                       partial region gets float atomics into the zeroed dw: not bit-reproducible).
#pragma once
#include "common.h"
#include "conv.h"
#include "igemm.h"  // f32x4, FastDiv

#define WD_THREADS 256
#define WD_ST 17  // LDS row stride (floats)

template <bool BF>
__device__ __forceinline__ void wd_mfma_step(const float* As, const float* Bs, int arow0, int brow0, int li, int lk,
                                             f32x4 (&acc)[2][2]) {
    if constexpr (BF) {  // (igemm.h: bf16 compute mode) one MFMA per 16-column step; this lane's k-values: lk*4 + 0..3
        s16x4 av[2], bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* ar = As + (arow0 + t * 16 + li) * WD_ST + lk * 4;
            const float* br = Bs + (brow0 + t * 16 + li) * WD_ST + lk * 4;
            av[t] = bf16_pack4(ar[0], ar[1], ar[2], ar[3]);
            bv[t] = bf16_pack4(br[0], br[1], br[2], br[3]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = MFMA_BF16(av[t], bv[u], acc[t][u]);
        return;
    }
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 4) {
        float av[2], bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            av[t] = As[(arow0 + t * 16 + li) * WD_ST + k4 + lk];
            bv[t] = Bs[(brow0 + t * 16 + li) * WD_ST + k4 + lk];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
    }
}

#define WD_KTAB_MAX 2304  // longest reduction (Cin x taps) the gather's LDS tap table holds

// CW = channel waves: the tile is 32*CW channels x 64 pixels (CW = 1 keeps layers with <= 32 output channels from
// computing a half-empty 64-channel tile).  Wave layout: CW x (4/CW) waves, each 2 channel tiles x CW pixel tiles.
template <bool TR, int CW, bool BF = false>
__global__ __launch_bounds__(WD_THREADS) void wide_gather_k(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B,
                                                            ConvGeom g, int act, float slope, int accumulate,
                                                            int kpad) {
    constexpr int NCH = 32 * CW;  // channels per tile
    constexpr int WJ = 2 * CW;    // weight elements a thread fills per K step
    constexpr int PXT = CW;       // 16-pixel tiles per wave
    __shared__ float Ws[2][NCH * WD_ST];  // [channel][k]
    __shared__ float Xs[2][64 * WD_ST];   // [pixel][k]
    extern __shared__ int wd_kt[];          // tap tables, `kpad` entries each (kpad = longest class reduction, padded to 16)
    int* const kt_delta = wd_kt;            // k -> offset of its tap from the pixel's base address
    int* const kt_hw = wd_kt + kpad;        // k -> (i_h << 16) | i_w, tap position for the bounds check
    int* const kt_w = wd_kt + 2 * kpad;     // k -> weight offset without the output channel (transposed layout only)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;

    // output-parity class (transposed) / the whole output (plain)
    int rh = 0, rw = 0, AH = g.OH, AW = g.OW, kh0 = 0, kw0 = 0, nkh = g.KH, nkw = g.KW, dh0 = 0, dw0 = 0;
    if (TR) {
        rh = blockIdx.z / g.sw; rw = blockIdx.z % g.sw;
        AH = (g.OH - rh + g.sh - 1) / g.sh; AW = (g.OW - rw + g.sw - 1) / g.sw;
        if (AH <= 0 || AW <= 0) return;
        kh0 = (rh + g.ph) % g.sh; kw0 = (rw + g.pw) % g.sw;
        nkh = kh0 < g.KH ? (g.KH - kh0 + g.sh - 1) / g.sh : 0;
        nkw = kw0 < g.KW ? (g.KW - kw0 + g.sw - 1) / g.sw : 0;
        dh0 = (rh + g.ph - kh0) / g.sh; dw0 = (rw + g.pw - kw0) / g.sw;  // exact: multiples of the stride
    }
    const int ntaps = nkh * nkw, K = g.Cin * ntaps, KK = g.KH * g.KW, HW = g.H * g.W;
    const int nsteps = (K + 15) / 16;
    const int64_t npix = (int64_t)B * AH * AW;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * NCH;
    if (p0 >= npix) return;

    // tap table; the padding entries up to the last K step fail every bounds check
    for (int k = tid; k < nsteps * 16; k += WD_THREADS) {
        int delta = 0, hw = 0x7fff0000, wo = 0;
        if (k < K) {
            const int ci = k / ntaps, ti = k - ci * ntaps;
            const int i_h = ti / nkw, i_w = ti - i_h * nkw;
            delta = TR ? ci * HW - i_h * g.W - i_w : ci * HW + i_h * g.W + i_w;
            hw = (i_h << 16) | i_w;
            if (TR) wo = ci * g.Cout * KK + (kh0 + i_h * g.sh) * g.KW + kw0 + i_w * g.sw;
        }
        kt_delta[k] = delta;
        kt_hw[k] = hw;
        if (TR) kt_w[k] = wo;
    }

    // this thread's pixel for the operand fill (64 pixels x 4 k-columns per thread and step)
    const int64_t p = p0 + (tid & 63);
    const bool pvalid = p < npix;
    int aw = 0, ah = 0;
    int64_t b = 0;
    if (pvalid) {
        aw = (int)(p % AW);
        const int64_t t = p / AW;
        ah = (int)(t % AH);
        b = t / AH;
    }
    // an invalid pixel gets a base row far outside the image, so its loads fail the bounds check like any halo tap
    const int ihb = !pvalid ? -0x100000 : (TR ? ah + dh0 : ah * g.sh - g.ph), iwb = TR ? aw + dw0 : aw * g.sw - g.pw;
    const float* xp = x + b * (int64_t)g.Cin * HW + (int64_t)ihb * g.W + iwb;  // only dereferenced in bounds
    const int wq = __builtin_amdgcn_readfirstlane(wid);
    __syncthreads();

    float xr[4], wr[WJ];
    auto load_step = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // activations: pixel = tid & 63, k column = wave + 4j (wave-uniform)
            const int k = k0 + wq + 4 * j;
            const int hw = kt_hw[k], delta = kt_delta[k];
            const int ih = TR ? ihb - (hw >> 16) : ihb + (hw >> 16), iw = TR ? iwb - (hw & 0xffff) : iwb + (hw & 0xffff);
            xr[j] = ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) ? xp[delta] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (!TR) {  // [n][K] row-major: k fastest over the lanes
                const int k = k0 + (tid & 15), n = n0 + (tid >> 4) + 16 * j;
                wr[j] = (n < g.Cout && k < K) ? w[(int64_t)n * K + k] : 0.f;
            } else {    // [ci][n][kh][kw]: n fastest over the lanes
                const int k = k0 + tid / NCH + (WD_THREADS / NCH) * j, n = n0 + tid % NCH;
                wr[j] = (n < g.Cout && k < K) ? w[kt_w[k] + n * KK] : 0.f;
            }
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Xs[buf][(tid & 63) * WD_ST + wq + 4 * j] = xr[j];
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            if (!TR) Ws[buf][((tid >> 4) + 16 * j) * WD_ST + (tid & 15)] = wr[j];
            else Ws[buf][(tid % NCH) * WD_ST + tid / NCH + (WD_THREADS / NCH) * j] = wr[j];
        }
    };

    f32x4 acc[2][PXT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < PXT; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ch0 = (CW == 2 ? (wid >> 1) : 0) * 32, px0 = (CW == 2 ? (wid & 1) * 32 : wid * 16);
    if (nsteps > 0) {
        load_step(0);
        store_step(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_step((s + 1) * 16);  // in flight during the MFMAs below
        const float* Wb = Ws[s & 1];
        const float* Xb = Xs[s & 1];
        if constexpr (BF) {
            s16x4 av[2], bv[PXT];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float* r = Wb + (ch0 + t * 16 + li) * WD_ST + lk * 4;
                av[t] = bf16_pack4(r[0], r[1], r[2], r[3]);
            }
#pragma unroll
            for (int u = 0; u < PXT; ++u) {
                const float* r = Xb + (px0 + u * 16 + li) * WD_ST + lk * 4;
                bv[u] = bf16_pack4(r[0], r[1], r[2], r[3]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < PXT; ++u) acc[t][u] = MFMA_BF16(av[t], bv[u], acc[t][u]);
        } else
#pragma unroll
        for (int k4 = 0; k4 < 16; k4 += 4) {
            float av[2], bv[PXT];
#pragma unroll
            for (int t = 0; t < 2; ++t) av[t] = Wb[(ch0 + t * 16 + li) * WD_ST + k4 + lk];
#pragma unroll
            for (int u = 0; u < PXT; ++u) bv[u] = Xb[(px0 + u * 16 + li) * WD_ST + k4 + lk];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < PXT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
        }
        if (s + 1 < nsteps) store_step((s + 1) & 1);
        __syncthreads();
    }

    // D: row (channel) = lk*4 + r, col (pixel) = li
    const int64_t ohw = (int64_t)g.OH * g.OW;
#pragma unroll
    for (int u = 0; u < PXT; ++u) {
        const int64_t q = p0 + px0 + u * 16 + li;
        if (q >= npix) continue;
        const int qaw = (int)(q % AW);
        const int64_t t = q / AW;
        const int qah = (int)(t % AH);
        const int64_t qb = t / AH;
        const int oh = TR ? qah * g.sh + rh : qah, ow = TR ? qaw * g.sw + rw : qaw;
        float* yp = y + qb * g.Cout * ohw + (int64_t)oh * g.OW + ow;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + ch0 + t2 * 16 + lk * 4 + r;
                if (n < g.Cout) {
                    float v = act_apply(acc[t2][u][r] + (bias != nullptr ? bias[n] : 0.f), act, slope);
                    if (accumulate) v += yp[n * ohw];
                    yp[n * ohw] = v;
                }
            }
    }
}

struct WideWgArgs {
    const float* S;  // [B][A][SH][SW]
    const float* G;  // [B][C][GH][GW]
    float* dw;       // [A][C][KH][KW]
    float* part;     // [grid.x][A][C][KH][KW] partial sums (reduced by wide_wgrad_reduce_k), or nullptr: float atomics into dw
    int B, A, SH, SW, C, GH, GW, KH, KW, sh, sw, ph, pw;
    int64_t R, chunk;       // reduction length B*SH*SW and the slice of it one grid.x index takes
    FastDiv d_SHW, d_SW, d_KK, d_KW;
};

template <bool BF>
__global__ __launch_bounds__(WD_THREADS) void wide_wgrad_k(WideWgArgs a) {
    __shared__ float Ss[2][64 * WD_ST];  // [a][pixel]
    __shared__ float Gs[2][64 * WD_ST];  // [(c,tap)][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int SHW = a.SH * a.SW, KK = a.KH * a.KW, NK = a.C * KK;
    const int a0 = blockIdx.y * 64, n0 = blockIdx.z * 64;
    const int64_t r_begin = (int64_t)blockIdx.x * a.chunk;
    const int64_t r_end = r_begin + a.chunk < a.R ? r_begin + a.chunk : a.R;
    if (r_begin >= r_end) return;

    // per-thread constants of the G fill: (c, kh, kw) of its four columns
    int gc[4], gkh[4], gkw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + (tid >> 4) + 16 * j;
        gc[j] = -1;
        gkh[j] = gkw[j] = 0;
        if (n < NK) {
            gc[j] = fdiv((uint32_t)n, a.d_KK);
            const int tap = n - gc[j] * KK;
            gkh[j] = fdiv((uint32_t)tap, a.d_KW);
            gkw[j] = tap - gkh[j] * a.KW;
        }
    }
    float sr[4], gr[4];
    auto load_step = [&](int64_t r0) {
        const int64_t r = r0 + (tid & 15);  // this thread's pixel in the step (16 pixels, pixel fastest over lanes)
        const bool rv = r < r_end;
        int b = 0, oh = 0, ow = 0, pix = 0;
        if (rv) {
            b = fdiv((uint32_t)r, a.d_SHW);
            pix = (int)r - b * SHW;
            oh = fdiv((uint32_t)pix, a.d_SW);
            ow = pix - oh * a.SW;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int aa = a0 + (tid >> 4) + 16 * j;
            sr[j] = (rv && aa < a.A) ? a.S[((int64_t)b * a.A + aa) * SHW + pix] : 0.f;
            float v = 0.f;
            if (rv && gc[j] >= 0) {
                const int gh = oh * a.sh - a.ph + gkh[j], gw = ow * a.sw - a.pw + gkw[j];
                if (gh >= 0 && gh < a.GH && gw >= 0 && gw < a.GW) v = a.G[(((int64_t)b * a.C + gc[j]) * a.GH + gh) * a.GW + gw];
            }
            gr[j] = v;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Ss[buf][((tid >> 4) + 16 * j) * WD_ST + (tid & 15)] = sr[j];
            Gs[buf][((tid >> 4) + 16 * j) * WD_ST + (tid & 15)] = gr[j];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_half = wid >> 1, n_half = wid & 1;
    const int nsteps = (int)((r_end - r_begin + 15) / 16);
    load_step(r_begin);
    store_step(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_step(r_begin + (int64_t)(s + 1) * 16);
        wd_mfma_step<BF>(Ss[s & 1], Gs[s & 1], a_half * 32, n_half * 32, li, lk, acc);
        if (s + 1 < nsteps) store_step((s + 1) & 1);
        __syncthreads();
    }
    // D: row (a) = lk*4 + r, col ((c,tap)) = li
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int aa = a0 + a_half * 32 + t * 16 + lk * 4 + r;
                const int n = n0 + n_half * 32 + u * 16 + li;
                if (aa < a.A && n < NK) {
                    if (a.part != nullptr) a.part[((int64_t)blockIdx.x * a.A + aa) * NK + n] = acc[t][u][r];
                    else atomicAdd(&a.dw[(int64_t)aa * NK + n], acc[t][u][r]);
                }
            }
}

// dw[e] += sum over rows of part[row][e]: 64 elements x 4 row lanes per workgroup, four loads in flight per lane.
// One workgroup owns an element and adds its rows in a fixed order: bit-identical from run to run (no float atomics).
__global__ __launch_bounds__(256) void wide_wgrad_reduce_k(const float* __restrict__ part, float* __restrict__ dw,
                                                           int64_t n, int rows) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int r = ty;
        for (; r + 12 < rows; r += 16) {
            s0 += part[(int64_t)r * n + e];
            s1 += part[(int64_t)(r + 4) * n + e];
            s2 += part[(int64_t)(r + 8) * n + e];
            s3 += part[(int64_t)(r + 12) * n + e];
        }
        for (; r < rows; r += 4) s0 += part[(int64_t)r * n + e];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && e < n) dw[e] += (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
