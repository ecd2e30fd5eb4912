// conv.h - direct (gather-form) convolution kernels, NCHW fp32.
//
// Channel counts on this path are tiny (1..32 for the notebook model, 3..24 for the 1-D
// models), so the round-1 kernels keep one output pixel per lane with a small tile of
// output channels in registers; the weight index is wave-uniform, so weights come through
// the scalar cache (s_load) and each vector load of x feeds COT FMAs.
//
//   conv_gather_k   : y[b,co,oh,ow] = act(bias[co] + sum_{ci,kh,kw} x[b,ci,oh*sh-ph+kh,ow*sw-pw+kw] * w[co][ci][kh][kw])
//                     = Conv forward, and ConvTranspose input-gradient.
//   convT_gather_k  : y[b,co,oh,ow] = act(bias[co] + sum_{ci,kh,kw : (oh+ph-kh)%sh==0 ...} x[b,ci,(oh+ph-kh)/sh,(ow+pw-kw)/sw] * w[ci][co][kh][kw])
//                     = ConvTranspose forward, and Conv input-gradient.  One launch z-slice per
//                     output parity class so the tap loops are uniform for the whole block.
//   conv_wgrad_k    : dw[a][c][kh][kw] += sum_{b,oh,ow} S[b,a,oh,ow] * G[b,c,oh*sh-ph+kh,ow*sw-pw+kw]
//                     (Conv: S=dy, G=x;  ConvT: S=x, G=dy; same [a][c] index order as the weight).
#pragma once
#include "common.h"

struct ConvGeom {
    int Cin, H, W, Cout, OH, OW, KH, KW, sh, sw, ph, pw;
};

template <int COT>
__global__ __launch_bounds__(256) void conv_gather_k(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     int B, ConvGeom g, int act, float slope, int accumulate) {
    const int64_t npix = (int64_t)B * g.OH * g.OW;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int ow = (int)(p % g.OW);
    const int64_t t = p / g.OW;
    const int oh = (int)(t % g.OH);
    const int64_t b = t / g.OH;
    const int co0 = blockIdx.y * COT;

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = (bias != nullptr && co0 + j < g.Cout) ? bias[co0 + j] : 0.f;

    const float* xb = x + b * (int64_t)g.Cin * g.H * g.W;
    const int wstride = g.Cin * g.KH * g.KW;  // between consecutive output channels
    const int ih0 = oh * g.sh - g.ph, iw0 = ow * g.sw - g.pw;
    for (int ci = 0; ci < g.Cin; ++ci) {
        for (int kh = 0; kh < g.KH; ++kh) {
            const int ih = ih0 + kh;
            const bool hok = (ih >= 0) && (ih < g.H);
            for (int kw = 0; kw < g.KW; ++kw) {
                const int iw = iw0 + kw;
                const bool ok = hok && (iw >= 0) && (iw < g.W);
                const float xv = ok ? xb[((int64_t)ci * g.H + ih) * g.W + iw] : 0.f;
                const float* wp = w + ((int64_t)co0 * g.Cin + ci) * g.KH * g.KW + kh * g.KW + kw;
#pragma unroll
                for (int j = 0; j < COT; ++j)
                    if (co0 + j < g.Cout) acc[j] = fmaf(xv, wp[(int64_t)j * wstride], acc[j]);
            }
        }
    }
    const int64_t ohw = (int64_t)g.OH * g.OW;
    float* yb = y + (b * g.Cout + co0) * ohw + (int64_t)oh * g.OW + ow;
#pragma unroll
    for (int j = 0; j < COT; ++j)
        if (co0 + j < g.Cout) {
            float v = act_apply(acc[j], act, slope);
            if (accumulate) v += yb[j * ohw];
            yb[j * ohw] = v;
        }
}

// Geometry here is that of the TRANSPOSED conv: input (Cin,H,W) -> output (Cout,OH,OW).
template <int COT>
__global__ __launch_bounds__(256) void convT_gather_k(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int B, ConvGeom g, int act, float slope, int accumulate) {
    // parity class of this z-slice
    const int rh = blockIdx.z / g.sw, rw = blockIdx.z % g.sw;
    const int AH = (g.OH - rh + g.sh - 1) / g.sh, AW = (g.OW - rw + g.sw - 1) / g.sw;  // outputs in class
    if (AH <= 0 || AW <= 0) return;
    const int64_t npix = (int64_t)B * AH * AW;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const int aw = (int)(p % AW);
    const int64_t t = p / AW;
    const int ah = (int)(t % AH);
    const int64_t b = t / AH;
    const int oh = ah * g.sh + rh, ow = aw * g.sw + rw;
    const int co0 = blockIdx.y * COT;

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = (bias != nullptr && co0 + j < g.Cout) ? bias[co0 + j] : 0.f;

    const float* xb = x + b * (int64_t)g.Cin * g.H * g.W;
    const int kh_first = (rh + g.ph) % g.sh, kw_first = (rw + g.pw) % g.sw;
    const int khw = g.KH * g.KW;
    for (int ci = 0; ci < g.Cin; ++ci) {
        for (int kh = kh_first; kh < g.KH; kh += g.sh) {
            const int ih = (oh + g.ph - kh) / g.sh;  // exact: numerator is a multiple of sh
            const bool hok = (oh + g.ph - kh >= 0) && (ih < g.H);
            for (int kw = kw_first; kw < g.KW; kw += g.sw) {
                const int iw = (ow + g.pw - kw) / g.sw;
                const bool ok = hok && (ow + g.pw - kw >= 0) && (iw < g.W);
                const float xv = ok ? xb[((int64_t)ci * g.H + ih) * g.W + iw] : 0.f;
                const float* wp = w + ((int64_t)ci * g.Cout + co0) * khw + kh * g.KW + kw;
#pragma unroll
                for (int j = 0; j < COT; ++j)
                    if (co0 + j < g.Cout) acc[j] = fmaf(xv, wp[(int64_t)j * khw], acc[j]);
            }
        }
    }
    const int64_t ohw = (int64_t)g.OH * g.OW;
    float* yb = y + (b * g.Cout + co0) * ohw + (int64_t)oh * g.OW + ow;
#pragma unroll
    for (int j = 0; j < COT; ++j)
        if (co0 + j < g.Cout) {
            float v = act_apply(acc[j], act, slope);
            if (accumulate) v += yb[j * ohw];
            yb[j * ohw] = v;
        }
}

// Weight gradient.  S: [B][A][SH][SW] (small grid), G: [B][C][GH][GW] (big grid).
// grid.x = chunks of the (b,oh,ow) reduction, grid.y = a-tile * C + c, grid.z = tap group.
// Each lane accumulates AT x TG partial sums over its pixels; wave shuffle + LDS reduce;
// one atomicAdd per (block, element) into dw (dw is zeroed once per step by the executor).  Float atomics: the
// gradients of a layer that falls back to this kernel are NOT bit-reproducible from run to run (none of the benched
// models has such a layer; DESIGN.md, data layout).
template <int AT, int TG>
__global__ __launch_bounds__(256) void conv_wgrad_k(const float* __restrict__ S, const float* __restrict__ G,
                                                    float* __restrict__ dw, int B, int A, int SH, int SW, int C,
                                                    int GH, int GW, int KH, int KW, int sh, int sw, int ph, int pw,
                                                    int64_t chunk) {
    __shared__ float red[4][AT * TG];
    const int c = blockIdx.y % C;
    const int a0 = (blockIdx.y / C) * AT;
    const int tap0 = blockIdx.z * TG;
    const int ntap = KH * KW;
    const int64_t R = (int64_t)B * SH * SW;
    const int64_t r0 = (int64_t)blockIdx.x * chunk;
    const int64_t r1 = (r0 + chunk < R) ? r0 + chunk : R;

    float acc[AT][TG];
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int t = 0; t < TG; ++t) acc[i][t] = 0.f;

    const int64_t shw = (int64_t)SH * SW, ghw = (int64_t)GH * GW;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const int ow = (int)(r % SW);
        const int64_t q = r / SW;
        const int oh = (int)(q % SH);
        const int64_t b = q / SH;
        float s[AT];
#pragma unroll
        for (int i = 0; i < AT; ++i) s[i] = (a0 + i < A) ? S[(b * A + a0 + i) * shw + (int64_t)oh * SW + ow] : 0.f;
        const float* gb = G + (b * C + c) * ghw;
        const int ih0 = oh * sh - ph, iw0 = ow * sw - pw;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const int tap = tap0 + t;
            const int kh = tap / KW, kw = tap - kh * KW;
            const int ih = ih0 + kh, iw = iw0 + kw;
            const bool ok = (tap < ntap) && ih >= 0 && ih < GH && iw >= 0 && iw < GW;
            const float gv = ok ? gb[(int64_t)ih * GW + iw] : 0.f;
#pragma unroll
            for (int i = 0; i < AT; ++i) acc[i][t] = fmaf(s[i], gv, acc[i][t]);
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const float v = wave_sum(acc[i][t]);
            if (lane == 0) red[wid][i * TG + t] = v;
        }
    __syncthreads();
    if (threadIdx.x < AT * TG) {
        const int i = threadIdx.x / TG, t = threadIdx.x % TG;
        const int tap = tap0 + t;
        if (a0 + i < A && tap < ntap) {
            const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            atomicAdd(&dw[((int64_t)(a0 + i) * C + c) * ntap + tap], v);
        }
    }
}
