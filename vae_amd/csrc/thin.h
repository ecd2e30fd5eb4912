// thin.h - streaming kernels for the outermost layers of a Conv2d VAE: the first convolution (1 input channel) and
// the last transposed convolution (1 output channel), both 3x3, stride 2, padding 1 (notebook model:
// vae-checkpoint.ipynb:31-52 conv1 1->8, :84-104 convT3 8->1 with output_padding 1).
//
// These two layers move most of the step's bytes (the 28x28 images and the 8x14x14 maps next to them) and do almost
// no arithmetic (9 or 18 multiply-adds per value), so the chunked LDS/MFMA machinery of igemm.h only costs them:
// 5-9 us of table set-up per workgroup and a staged copy of data that is used once.  Here a 256-thread workgroup
// walks whole samples: the single-channel image of a sample goes through a small zero-haloed LDS tile (it is read 9
// times by neighbouring lanes), every thread owns one pixel of the 8-channel side and reads / writes it straight
// from / to global memory (coalesced per channel), the 72 weights are wave-uniform scalar loads, and weight-gradient
// products run on the matrix pipe (th_outer_mfma) and accumulate over all samples of the workgroup.  There is no
// set-up phase, the per-element code is branch-free (modes folded into coefficients, activation a template argument),
// and the loads of the samples TWO rounds ahead are in flight in a second register set: a round is ~0.5 us of
// compute, a global load under the kernel's own traffic takes 2-3 us (one round ahead: 32 us, two: 12 us for the
// transposed forward).  Measured at B = 4096 against the chunked kernels they replace: conv1 forward 21 -> 12 us,
// conv1 weight gradient 26 -> 18 us, convT3 forward 36 -> 12 us (51 MB: 4.1 TB/s), convT3 input + weight gradient
// 33 + 26 -> 25 us (77 MB: 3.1 TB/s).
//
//   thin_conv_fwd_k<C>    y[s][co][oh][ow] = act(b[co] + sum_k x[s][2oh-1+kh][2ow-1+kw] w[co][k])          (Cin = 1)
//   thin_conv_wgrad_k<C>  dw[co][k] = sum g'[s][co][oh][ow] x[...], db[co] = sum g'                          (first layer: no dx)
//   thin_convT_fwd_k<C>   y[s][2a+rh][2b+rw] = act(b + sum_ci sum_taps in'[s][ci][a+dh][b+dw] w[ci][k])       (Cout = 1)
//   thin_convT_bwd_k<C>   din[s][ci][a][b] = mask * sum_k g'[s][2a-1+kh][2b-1+kw] w[ci][k], the producer BatchNorm's
//                         backward sums, dw[ci][k] = sum in' g', db = sum g'
// in' = the layer input as the forward saw it (producer BatchNorm + ReLU applied on load when it is virtual), g' = the
// gradient of the pre-activation (stored gradient times act'(output)).  C = channels of the wide side (<= 8).
#pragma once
#include "common.h"
#include "igemm.h"  // TensorIn, tensor_transform, FastDiv

#define TH_THREADS 256
#define TH_MAX_HW 256  // pixels of the wide side per sample (one thread each)

struct ThinArgs {
    TensorIn in;         // fwd: layer input.  bwd/wgrad: layer input as the forward saw it
    TensorIn g;          // bwd/wgrad: d(pre-activation) of this layer = transform(dout, out)
    const float* w;      // conv [C][1][3][3] / convT [C][1][3][3]: w[c * 9 + kh * 3 + kw] either way
    const float* bias;   // [C] (conv) or [1] (convT), may be null
    float* out;          // fwd: output.  convT bwd: din (may be null)
    float* slab;         // convT bwd: the producer BatchNorm's backward sums, one row [C][4] per workgroup (or null)
    const float* ep_coef;  // convT bwd: that BatchNorm's save block (mean, rstd, scale, beta)
    int ep_relu;
    float* part;         // wgrad / bwd: [grid][9 * C + C] (conv) or [grid][9 * C + 1] (convT) partial rows
    int want_bias;
    int B, H, W;         // H x W = the wide side (conv: output 14x14, convT: input 14x14); the image is 2H x 2W
    int act;
    float slope;
    FastDiv d_W, d_W4;   // by W and by 2W / 4 (float4 groups per image row)
    BnFin fin;           // convT fwd: the producer BatchNorm is finalised in this kernel's prologue (C > 0; igemm.h)
    float* acc_save;     // convT bwd: the backward sums also go to the accumulators behind this save block (or null)
};

// Branch-free helpers: the modes are folded into coefficients (identity when a transform is absent) and the
// activation is a template argument, so the per-element code has no mode branches (with run-time modes the compiler
// emitted a branch and a wait per element and the kernels ran no faster than the chunked ones).
template <int ACT>
__device__ __forceinline__ float th_act(float v, float slope) {
    if (ACT == VAE_ACT_LEAKY) return v > 0.f ? v : v * slope;
    if (ACT == VAE_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-v));  // (1 ulp reciprocal)
    return v;
}
template <int ACT>
__device__ __forceinline__ float th_act_grad(float y, float slope) {  // d(act)/d(pre-activation) from the OUTPUT
    if (ACT == VAE_ACT_LEAKY) return y > 0.f ? 1.f : slope;
    if (ACT == VAE_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}
// Input transform coefficients in LDS: [0,C) mean, [C,2C) rstd, [2C,3C) scale, [3C,4C) beta, [4C] ReLU threshold
// (0 with ReLU, -inf without).  Plain input: mean 0, rstd 0, scale 1, beta 0.
template <int C>
__device__ __forceinline__ void th_in_coef(const TensorIn& t, float* coef) {
    for (int e = threadIdx.x; e < 4 * C; e += TH_THREADS) {
        float v = (e >= 2 * C && e < 3 * C) ? 1.f : 0.f;
        if (t.mode == TR_BN) v = t.coef[e];
        coef[e] = v;
    }
    if (threadIdx.x == 0) coef[4 * C] = (t.mode == TR_BN && t.relu) ? 0.f : -__builtin_inff();
}
// Gradient transform coefficients in LDS: g' = (ga*g + gb*(y - mean) + gc) * act'(y): [0,C) mean, [C,2C) ga, [2C,3C) gb,
// [3C,4C) gc.  Without a BatchNorm behind the layer: mean 0, ga 1, gb 0, gc 0.
template <int C>
__device__ __forceinline__ void th_grad_coef(const TensorIn& t, float* coef) {
    for (int e = threadIdx.x; e < 4 * C; e += TH_THREADS) {
        const int q = e / C, c = e - q * C;
        float v = q == 1 ? 1.f : 0.f;
        if (t.mode == TR_GRAD_BN) v = t.coef[(q == 0 ? 0 : 3 + q) * C + c];  // save block: mean, ..., a (4), b (5), c (6)
        coef[e] = v;
    }
}

// Sum of `v` over the workgroup for NV values per thread: wave DPP sums, then the four waves through LDS.  red: >= 4*NV
// floats.  The result lands in out[0..NV) (global or LDS) written by thread j of the block.
template <int NV>
__device__ __forceinline__ void th_block_sums(float (&v)[NV], float* red, float* out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float t = wave_sum(v[j]);
        if (lane == 0) red[wid * NV + j] = t;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < NV; j += TH_THREADS) out[j] = (red[j] + red[NV + j]) + (red[2 * NV + j] + red[3 * NV + j]);
    __syncthreads();
}

// Weight-gradient products on the matrix pipe: dw[c][k] += sum over this wave's 64 pixels of a[c] * b[k] is a
// [16 x 64] x [64 x 16] product (rows c < C and columns k < 9 are the real ones).  Every lane leaves its pixel's
// a[0..C) and b[0..9) in its own 17-float row of `mm` (the wave's 64 rows; stride 17: conflict-free), and sixteen
// v_mfma_f32_16x16x4_f32 walk the rows: four accumulator registers instead of 9*C, and no per-product wave sums at the
// end.  Only the wave's own rows are read, so no workgroup barrier is needed (a wave's LDS accesses execute in order).
// Rows >= C / columns >= 9 of the result are meaningless (they multiply the neighbouring values) and never read.
#define TH_MM_STRIDE 17
template <int C>
__device__ __forceinline__ void th_outer_mfma(float* mm, const float (&av)[C], const float (&bv)[9], f32x4& acc) {
    static_assert(C <= 8, "a row of mm holds 8 + 9 values");
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    float* row = mm + lane * TH_MM_STRIDE;
#pragma unroll
    for (int c = 0; c < C; ++c) row[c] = av[c];
#pragma unroll
    for (int k = 0; k < 9; ++k) row[8 + k] = bv[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float* p = mm + lk * TH_MM_STRIDE + li;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p[4 * i * TH_MM_STRIDE], p[4 * i * TH_MM_STRIDE + 8], acc, 0, 0, 0);
    asm volatile("" ::: "memory");
}
// The four waves' accumulators -> out[c * 9 + k] (fixed order).  red: >= 4 * 256 floats.
template <int C>
__device__ __forceinline__ void th_outer_reduce(const f32x4& acc, float* red, float* out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wid * 16 + lk * 4 + r) * 16 + li] = acc[r];  // D: row = lk*4 + r, column = li
    __syncthreads();
    for (int j = threadIdx.x; j < 9 * C; j += TH_THREADS) {
        const int c = j / 9, k = j - c * 9;
        out[j] = (red[c * 16 + k] + red[(16 + c) * 16 + k]) + (red[(32 + c) * 16 + k] + red[(48 + c) * 16 + k]);
    }
}

// The single-channel 2H x 2W image of a sample goes into a zero-haloed tile: element (r, c) at tile[(r + 1) * ws + c + 1]
// (row 0 and column 0 are the halo of the padding = 1 low side; the high side never reaches past 2H - 1).  Loading and
// storing are split so that the NEXT sample's global loads are in flight while the current sample is computed:
// th_img_load issues them into registers (one float4 per thread: the host checks 4 | 2W and 2H * 2W <= 1024),
// th_img_store writes the tile (GACT >= 0: the image is a gradient, times act'(y) on the way).
struct ThImg {
    float4 v, y;
};
template <int GACT>
__device__ __forceinline__ void th_img_load(const float* src, const float* src2, int64_t base, int n4, ThImg& r) {
    const int e = threadIdx.x < n4 ? threadIdx.x : 0;  // (clamped: branch-free, the extra lanes re-read element 0)
    r.v = reinterpret_cast<const float4*>(src + base)[e];
    if (GACT > VAE_ACT_NONE) r.y = reinterpret_cast<const float4*>(src2 + base)[e];
}
template <int GACT>
__device__ __forceinline__ void th_img_store(float* tile, int W2, int ws, int n4, float slope, FastDiv d_W4, const ThImg& r) {
    const int e = threadIdx.x;
    if (e < n4) {
        float4 v = r.v;
        if (GACT > VAE_ACT_NONE) {
            v.x *= th_act_grad<GACT>(r.y.x, slope); v.y *= th_act_grad<GACT>(r.y.y, slope);
            v.z *= th_act_grad<GACT>(r.y.z, slope); v.w *= th_act_grad<GACT>(r.y.w, slope);
        }
        const int row = (int)fdiv((uint32_t)e, d_W4), col = (e - row * (W2 >> 2)) << 2;
        float* d = tile + (row + 1) * ws + col + 1;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

// ---- first convolution, forward -------------------------------------------------------------------------------
template <int C, int ACT>
__global__ __launch_bounds__(TH_THREADS) void thin_conv_fwd_k(ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H2 = 2 * a.H, W2 = 2 * a.W, ws = W2 + 1, tile_n = (H2 + 1) * ws, n4 = (H2 * W2) >> 2;
    float* tile[2] = {smem, smem + tile_n};
    const int tid = threadIdx.x, HW = a.H * a.W;
    const int oh = (int)fdiv((uint32_t)tid, a.d_W), ow = tid - oh * a.W;
    const bool active = tid < HW;
    const int64_t img = (int64_t)H2 * W2;
    // two register sets: the loads of the samples two rounds ahead are in flight (one round is ~0.5 us of compute,
    // a global load under this kernel's own traffic takes 2-3 us)
    const int g = gridDim.x, last = a.B - 1;
    ThImg nxA, nxB;
    th_img_load<-1>(a.in.src, nullptr, blockIdx.x * img, n4, nxA);  // (grid <= B)
    th_img_load<-1>(a.in.src, nullptr, min((int)blockIdx.x + g, last) * img, n4, nxB);
    for (int e = tid; e < 2 * tile_n; e += TH_THREADS) smem[e] = 0.f;  // the halo stays zero for every sample
    float wv[C][9], bv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        bv[c] = a.bias != nullptr ? a.bias[c] : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[c][k] = a.w[c * 9 + k];  // wave-uniform: scalar loads
    }
    lds_barrier();
    th_img_store<-1>(tile[0], W2, ws, n4, 0.f, a.d_W4, nxA);
    th_img_load<-1>(a.in.src, nullptr, min((int)blockIdx.x + 2 * g, last) * img, n4, nxA);
    auto compute = [&](const float* t, int s) {
        if (!active || s > last) return;
        const float* p = t + (2 * oh) * ws + 2 * ow;  // tile(2oh-1+kh, 2ow-1+kw) = p[kh*ws + kw]
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = p[kh * ws + kw];
        float* o = a.out + (int64_t)s * C * HW + tid;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float acc = bv[c];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(xv[k], wv[c][k], acc);
            o[c * HW] = th_act<ACT>(acc, a.slope);
        }
    };
    // (sample indices past the end are clamped: a harmless re-read of the last sample, never computed)
    for (int s = blockIdx.x; s < a.B; s += 2 * g) {
        lds_barrier();  // tile 0 is complete, tile 1 is free
        compute(tile[0], s);
        th_img_store<-1>(tile[1], W2, ws, n4, 0.f, a.d_W4, nxB);
        th_img_load<-1>(a.in.src, nullptr, min(s + 3 * g, last) * img, n4, nxB);
        lds_barrier();
        compute(tile[1], s + g);
        th_img_store<-1>(tile[0], W2, ws, n4, 0.f, a.d_W4, nxA);
        th_img_load<-1>(a.in.src, nullptr, min(s + 4 * g, last) * img, n4, nxA);
    }
}

// ---- first convolution, weight + bias gradient ------------------------------------------------------------------
template <int C, int ACT>
__global__ __launch_bounds__(TH_THREADS) void thin_conv_wgrad_k(ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H2 = 2 * a.H, W2 = 2 * a.W, ws = W2 + 1, tile_n = (H2 + 1) * ws, n4 = (H2 * W2) >> 2;
    float* tile[2] = {smem, smem + tile_n};
    float* coef = smem + 2 * tile_n;        // [4 * C + 1] the gradient transform's coefficients
    float* red = coef + 4 * C + 4;          // [1024]
    float* mm = red + 1024;                 // [256 * 17 + 16]
    const int tid = threadIdx.x, HW = a.H * a.W;
    const int oh = (int)fdiv((uint32_t)tid, a.d_W), ow = tid - oh * a.W;
    const bool active = tid < HW;
    const int own = active ? tid : 0;
    const int64_t img = (int64_t)H2 * W2;
    const int g = gridDim.x, last = a.B - 1;
    float* const mmw = mm + (tid >> 6) * 64 * TH_MM_STRIDE;
    ThImg nxA, nxB;
    float gA[C], yA[C], gB[C], yB[C];
    auto load_own = [&](int s, float (&gn)[C], float (&yn)[C]) {  // this thread's pixel of every channel of the gradient and the output
        const int64_t o = (int64_t)s * C * HW + own;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            gn[c] = a.g.src[o + c * HW];
            yn[c] = a.g.src2[o + c * HW];
        }
    };
    th_img_load<-1>(a.in.src, nullptr, blockIdx.x * img, n4, nxA);
    load_own(blockIdx.x, gA, yA);
    th_img_load<-1>(a.in.src, nullptr, min((int)blockIdx.x + g, last) * img, n4, nxB);
    load_own(min((int)blockIdx.x + g, last), gB, yB);
    for (int e = tid; e < 2 * tile_n; e += TH_THREADS) smem[e] = 0.f;
    th_grad_coef<C>(a.g, coef);
    f32x4 accw = {0.f, 0.f, 0.f, 0.f};  // this wave's [c][k] products
    float accb[C];                      // bias sums
#pragma unroll
    for (int c = 0; c < C; ++c) accb[c] = 0.f;
    lds_barrier();
    th_img_store<-1>(tile[0], W2, ws, n4, 0.f, a.d_W4, nxA);
    th_img_load<-1>(a.in.src, nullptr, min((int)blockIdx.x + 2 * g, last) * img, n4, nxA);
    // one round: the sample whose image is in `t` and whose own-pixel values are in (gn, yn); then those registers
    // are reloaded with the sample two rounds ahead
    auto round = [&](const float* t, int s, float (&gn)[C], float (&yn)[C]) {
        const bool on = active && s <= last;
        float gv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float q = fmaf(coef[C + c], gn[c], fmaf(coef[2 * C + c], yn[c] - coef[c], coef[3 * C + c]));
            gv[c] = on ? q * th_act_grad<ACT>(yn[c], a.slope) : 0.f;
        }
        load_own(min(s + 2 * g, last), gn, yn);
        const float* p = t + (active ? (2 * oh) * ws + 2 * ow : 0);
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = p[kh * ws + kw];
#pragma unroll
        for (int c = 0; c < C; ++c) accb[c] += gv[c];
        th_outer_mfma<C>(mmw, gv, xv, accw);  // (idle lanes: gv = 0)
    };
    for (int s = blockIdx.x; s < a.B; s += 2 * g) {
        lds_barrier();
        round(tile[0], s, gA, yA);
        th_img_store<-1>(tile[1], W2, ws, n4, 0.f, a.d_W4, nxB);
        th_img_load<-1>(a.in.src, nullptr, min(s + 3 * g, last) * img, n4, nxB);
        lds_barrier();
        round(tile[1], s + g, gB, yB);
        th_img_store<-1>(tile[0], W2, ws, n4, 0.f, a.d_W4, nxA);
        th_img_load<-1>(a.in.src, nullptr, min(s + 4 * g, last) * img, n4, nxA);
    }
    float* row = a.part + (int64_t)blockIdx.x * 10 * C;
    th_outer_reduce<C>(accw, red, row);
    th_block_sums<C>(accb, red, row + 9 * C);
}

// ---- last transposed convolution, forward ---------------------------------------------------------------------
// Thread (a, b) of the C-channel H x W input owns the 2 x 2 output quad (2a + rh, 2b + rw):
//   y(2a  , 2b  ) = i00 w11                       y(2a  , 2b+1) = i01 w10 + i00 w12
//   y(2a+1, 2b  ) = i10 w01 + i00 w21             y(2a+1, 2b+1) = i11 w00 + i10 w02 + i01 w20 + i00 w22
// with iJK = in'(a + J, b + K) (zero past the edge) and wKL = w[ci][kh = K][kw = L]: oh = 2 ih - 1 + kh.
template <int C, int ACT>
__global__ __launch_bounds__(TH_THREADS) void thin_convT_fwd_k(ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ws = a.W + 1, plane = (a.H + 1) * ws, tile_n = C * plane;  // zero row / column on the HIGH side
    float* tile[2] = {smem, smem + tile_n};
    float* coef = smem + 2 * tile_n;  // [4 * C + 1] producer BatchNorm
    const int tid = threadIdx.x, HW = a.H * a.W;
    const int ah = (int)fdiv((uint32_t)tid, a.d_W), aw = tid - ah * a.W;
    const bool active = tid < HW;
    const int own = active ? tid : 0;
    const int g = gridDim.x, last = a.B - 1;
    float vA[C], vB[C];  // two register sets: loads two rounds ahead
    auto load_own = [&](int s, float (&vn)[C]) {
        const int64_t o = (int64_t)s * C * HW + own;
#pragma unroll
        for (int c = 0; c < C; ++c) vn[c] = a.in.src[o + c * HW];
    };
    auto store_own = [&](float* t, const float (&vn)[C]) {  // transform (producer BatchNorm + ReLU) -> this thread's pixel of every plane
        if (!active) return;
        const float thr = coef[4 * C];
#pragma unroll
        for (int c = 0; c < C; ++c)
            t[c * plane + ah * ws + aw] = fmaxf(fmaf(vn[c] - coef[c], coef[2 * C + c], coef[3 * C + c]), thr);
    };
    BnFinRegs finr;
    bnfin_issue<false>(a.fin, finr);
    load_own(blockIdx.x, vA);
    load_own(min((int)blockIdx.x + g, last), vB);
    for (int e = tid; e < 2 * tile_n; e += TH_THREADS) smem[e] = 0.f;
    if (a.fin.C > 0) {  // every workgroup finalises the producer BatchNorm from its accumulators (requested above)
        bnfin_forward<TH_THREADS>(a.fin, finr, coef, coef + 4 * C + 4);
        if (tid == 0) coef[4 * C] = a.in.relu ? 0.f : -__builtin_inff();
    } else {
        th_in_coef<C>(a.in, coef);
    }
    float wv[C][9];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[c][k] = a.w[c * 9 + k];
    const float b0 = a.bias != nullptr ? a.bias[0] : 0.f;
    lds_barrier();
    store_own(tile[0], vA);
    load_own(min((int)blockIdx.x + 2 * g, last), vA);
    const int W2 = 2 * a.W;
    auto compute = [&](const float* t, int s) {
        if (active && s <= last) {
            float y00 = b0, y01 = b0, y10 = b0, y11 = b0;
            const float* p = t + ah * ws + aw;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float i00 = p[c * plane], i01 = p[c * plane + 1], i10 = p[c * plane + ws], i11 = p[c * plane + ws + 1];
                y00 = fmaf(i00, wv[c][4], y00);
                y01 = fmaf(i01, wv[c][3], fmaf(i00, wv[c][5], y01));
                y10 = fmaf(i10, wv[c][1], fmaf(i00, wv[c][7], y10));
                y11 = fmaf(i11, wv[c][0], fmaf(i10, wv[c][2], fmaf(i01, wv[c][6], fmaf(i00, wv[c][8], y11))));
            }
            float* o = a.out + (int64_t)s * 4 * HW + (2 * ah) * W2 + 2 * aw;
            *reinterpret_cast<float2*>(o) = make_float2(th_act<ACT>(y00, a.slope), th_act<ACT>(y01, a.slope));
            *reinterpret_cast<float2*>(o + W2) = make_float2(th_act<ACT>(y10, a.slope), th_act<ACT>(y11, a.slope));
        }
    };
    for (int s = blockIdx.x; s < a.B; s += 2 * g) {
        lds_barrier();
        compute(tile[0], s);
        store_own(tile[1], vB);
        load_own(min(s + 3 * g, last), vB);
        lds_barrier();
        compute(tile[1], s + g);
        store_own(tile[0], vA);
        load_own(min(s + 4 * g, last), vA);
    }
}

// ---- last transposed convolution, backward: input gradient (+ producer BatchNorm sums), weight and bias gradient ----
template <int C, int ACT>
__global__ __launch_bounds__(TH_THREADS) void thin_convT_bwd_k(ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H2 = 2 * a.H, W2 = 2 * a.W, ws = W2 + 1, tile_n = (H2 + 1) * ws, n4 = (H2 * W2) >> 2;
    float* tile[2] = {smem, smem + tile_n};   // g' image, zero halo on the low side
    float* coef = smem + 2 * tile_n;          // [4 * C + 1] producer BatchNorm (input transform and epilogue)
    float* red = coef + 4 * C + 4;            // [1024]
    float* mm = red + 1024;                   // [256 * 17 + 16]
    float* const mmw = mm + (threadIdx.x >> 6) * 64 * TH_MM_STRIDE;
    const int tid = threadIdx.x, HW = a.H * a.W;
    const int ah = (int)fdiv((uint32_t)tid, a.d_W), aw = tid - ah * a.W;
    const bool active = tid < HW;
    const int own = active ? tid : 0;
    const int64_t img = (int64_t)H2 * W2;
    const int g = gridDim.x, last = a.B - 1;
    ThImg nxA, nxB;
    float yA[C], yB[C];
    auto load_own = [&](int s, float (&yn)[C]) {
        const int64_t o = (int64_t)s * C * HW + own;
#pragma unroll
        for (int c = 0; c < C; ++c) yn[c] = a.in.src[o + c * HW];
    };
    th_img_load<ACT>(a.g.src, a.g.src2, blockIdx.x * img, n4, nxA);
    load_own(blockIdx.x, yA);
    th_img_load<ACT>(a.g.src, a.g.src2, min((int)blockIdx.x + g, last) * img, n4, nxB);
    load_own(min((int)blockIdx.x + g, last), yB);
    for (int e = tid; e < 2 * tile_n; e += TH_THREADS) smem[e] = 0.f;
    th_in_coef<C>(a.in, coef);
    float wv[C][9];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[c][k] = a.w[c * 9 + k];
    constexpr int NV = 2 * C + 1;  // [c] sum g, [c] sum g*xhat, bias
    float acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.f;
    f32x4 accw = {0.f, 0.f, 0.f, 0.f};  // this wave's [c][k] weight-gradient products
    lds_barrier();
    th_img_store<ACT>(tile[0], W2, ws, n4, a.slope, a.d_W4, nxA);
    th_img_load<ACT>(a.g.src, a.g.src2, min((int)blockIdx.x + 2 * g, last) * img, n4, nxA);
    auto round = [&](const float* t, int s, float (&yn)[C]) {
        float yin[C];
#pragma unroll
        for (int c = 0; c < C; ++c) yin[c] = yn[c];
        load_own(min(s + 2 * g, last), yn);
        const bool on = active && s <= last;
        {
            // g'(2a-1+kh, 2b-1+kw) = tile[(2a+kh) * ws + 2b+kw]   (idle threads read the corner and add nothing)
            const float* p = t + (active ? (2 * ah) * ws + 2 * aw : 0);
            const float live = on ? 1.f : 0.f;
            float gv[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) gv[kh * 3 + kw] = p[kh * ws + kw] * live;
            // the bias gradient is the sum of every g' of the sample: this thread's quad (2a+rh, 2b+rw) = gv[4,5,7,8]
            acc[NV - 1] += (gv[4] + gv[5]) + (gv[7] + gv[8]);
            const float thr = coef[4 * C];
            float* dx = a.out + (int64_t)s * C * HW + tid;
            float hv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float xc = yin[c] - coef[c];
                const float pre = fmaf(xc, coef[2 * C + c], coef[3 * C + c]);  // producer BatchNorm output
                hv[c] = fmaxf(pre, thr);                                       // the forward's input value
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) d = fmaf(gv[k], wv[c][k], d);
                // backward through the producer's (virtual) BatchNorm + ReLU: mask, and its two sums
                d = pre > thr ? d : 0.f;
                acc[c] += d;
                acc[C + c] = fmaf(d, xc * coef[C + c], acc[C + c]);
                if (a.out != nullptr && on) dx[c * HW] = d;
            }
            th_outer_mfma<C>(mmw, hv, gv, accw);  // dw[c][k] += in'[c] * g'[k]   (idle lanes: gv = 0)
        }
    };
    for (int s = blockIdx.x; s < a.B; s += 2 * g) {
        lds_barrier();
        round(tile[0], s, yA);
        th_img_store<ACT>(tile[1], W2, ws, n4, a.slope, a.d_W4, nxB);
        th_img_load<ACT>(a.g.src, a.g.src2, min(s + 3 * g, last) * img, n4, nxB);
        lds_barrier();
        round(tile[1], s + g, yB);
        th_img_store<ACT>(tile[0], W2, ws, n4, a.slope, a.d_W4, nxA);
        th_img_load<ACT>(a.g.src, a.g.src2, min(s + 4 * g, last) * img, n4, nxA);
    }
    float* row = a.part + (int64_t)blockIdx.x * (9 * C + 1);
    th_outer_reduce<C>(accw, red, row);
    __syncthreads();
    float* sums = red + 4 * NV;  // [NV] block totals
    th_block_sums<NV>(acc, red, sums);
    if (tid == 0) row[9 * C] = sums[NV - 1];
    if (a.slab != nullptr && tid < C) {
        float* sl = a.slab + ((int64_t)blockIdx.x * C + tid) * 4;
        sl[0] = sums[tid];
        sl[1] = sums[C + tid];
        sl[2] = 0.f;
        sl[3] = 0.f;
        if (a.acc_save != nullptr) bnacc_add_bwd(a.acc_save, C, tid, sums[tid], sums[C + tid]);
    }
}

// ================================================================================================================
// Mid-size layers (8 <-> 16 channels on 14x14 <-> 7x7 maps: conv2 / convT2 of the notebook model), same recipe: a
// 256-thread workgroup streams whole samples, loads two rounds ahead in a second register set, branch-free
// transforms, one LDS-only barrier per round, several workgroups per CU.
//
// mid_wgrad_k<MT, NT, S_GRAD, ACT>: weight + bias gradient
//   dw[a][c][kh][kw] = sum_{b,oh,ow} S[b,a,oh,ow] * G[b,c,oh*sh-ph+kh,ow*sw-pw+kw]
// (Conv: S = d(pre-activation) [Cout] on the small grid, G = layer input [Cin]; ConvT: S = layer input, G = d(pre-act)).
// Per sample the product is [Sc x P] x [P x Gc*KK] with P = SH*SW pixels as the MFMA's K: the four waves split the
// k-steps (4 pixels each), every wave keeps all MT x NT accumulator tiles for the whole kernel; A = S[a][pixel] from
// the LDS copy of S, B = G gathered from the zero-haloed LDS image at (pixel base of the k-step) + (lane constant
// (c, tap) offset); both index sets live in registers.  One cross-wave reduction per workgroup at the end; the partial
// row has the layout of igemm_wgrad_k's (accumulator-tile order, bias sums behind), so wgrad_finalize_k serves both.
// S_GRAD: S is the gradient-type tensor (then G is the input-type one), else the other way round.
struct MidWgArgs {
    TensorIn S, G;
    float* part;      // [grid][MT*NT*256 + 64]
    int want_bias;
    int B, SBR;       // SBR samples per round: SBR * SH * SW <= 64 pixels (the MFMA's K of a round)
    int Sc, SH, SW;   // S: channels, small grid
    int Gc, GH, GW;   // G: channels, big grid
    int KH, KW, sh, sw, ph, pw;
    int Hp, Wp;       // padded G image (halo = ph, pw on the low side)
    float slope;
    FastDiv dS_HW, dS_W, dS_CHW, dG_HW, dG_W, dG_CHW, d_KK, d_KW;
    BnFin fin;        // the BatchNorm behind the layer is finalised (backward) in this kernel's prologue (C > 0; igemm.h)
};
#define MID_KSW 4  // k-steps per wave: SBR * SH * SW <= 64

// NS / NG: staged elements per thread and round of S / G (SBR * Sc * SH * SW <= 256 * NS, likewise G).
template <int MT, int NT, int NS, int NG, bool S_GRAD, int ACT>
__global__ __launch_bounds__(TH_THREADS) void mid_wgrad_k(MidWgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int SHW = a.SH * a.SW, GHW = a.GH * a.GW, HpWp = a.Hp * a.Wp, KK = a.KH * a.KW, NK = a.Gc * KK;
    const int nS1 = a.Sc * SHW, nG1 = a.Gc * GHW, imgG1 = a.Gc * HpWp;  // per sample
    const int nS = a.SBR * nS1, nG = a.SBR * nG1, imgG = a.SBR * imgG1;  // per round
    const TensorIn& tin = S_GRAD ? a.G : a.S;   // the input-type tensor
    const TensorIn& tgr = S_GRAD ? a.S : a.G;   // the gradient-type tensor
    const int Cin_t = S_GRAD ? a.Gc : a.Sc, Cgr_t = S_GRAD ? a.Sc : a.Gc;
    // LDS: [2][nS] S tiles, [2][imgG] G images, input coefs [4*Cin_t + 4], gradient coefs [4*Cgr_t]; reduce area at the end
    float* lS[2] = {smem, smem + nS};
    float* lG[2] = {smem + 2 * nS, smem + 2 * nS + imgG};
    float* cin = smem + 2 * nS + 2 * imgG;
    float* cgr = cin + 4 * Cin_t + 4;
    const int g = gridDim.x;
    const int nrounds = (a.B + a.SBR - 1) / a.SBR, last = nrounds - 1;

    float sA[NS], sB[NS], sAy[S_GRAD ? NS : 1], sBy[S_GRAD ? NS : 1];
    float gA[NG], gB[NG], gAy[S_GRAD ? 1 : NG], gBy[S_GRAD ? 1 : NG];
    auto load = [&](int rd, float (&sv)[NS], float (&sy)[S_GRAD ? NS : 1], float (&gv)[NG], float (&gy)[S_GRAD ? 1 : NG]) {
        rd = min(rd, last);
        const int t0 = (int)opaque_tid();  // (keeps the slots' 64-bit addresses from being hoisted out of the round loop)
        const int sbr = min(a.SBR, a.B - rd * a.SBR);
        const float* ps = a.S.src + (int64_t)rd * nS;
        const float* pg = a.G.src + (int64_t)rd * nG;
        const float* ps2 = S_GRAD ? a.S.src2 + (int64_t)rd * nS : nullptr;
        const float* pg2 = S_GRAD ? nullptr : a.G.src2 + (int64_t)rd * nG;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int e = min(t0 + 256 * j, sbr * nS1 - 1);
            sv[j] = ps[e];
            if (S_GRAD) sy[j] = ps2[e];
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int e = min(t0 + 256 * j, sbr * nG1 - 1);
            gv[j] = pg[e];
            if (!S_GRAD) gy[j] = pg2[e];
        }
    };
    BnFinRegs finr;
    IG_TDECL;
    IG_T();
    bnfin_issue<true>(a.fin, finr);
    load(blockIdx.x, sA, sAy, gA, gAy);
    load(blockIdx.x + g, sB, sBy, gB, gBy);
    IG_T();

    // ---- once per workgroup
    for (int e = tid; e < 2 * imgG; e += TH_THREADS) lG[0][e] = 0.f;  // the halo stays zero
    for (int e = tid; e < 4 * Cin_t; e += TH_THREADS) {
        float v = (e >= 2 * Cin_t && e < 3 * Cin_t) ? 1.f : 0.f;
        if (tin.mode == TR_BN) v = tin.coef[e];
        cin[e] = v;
    }
    if (tid == 0) cin[4 * Cin_t] = (tin.mode == TR_BN && tin.relu) ? 0.f : -__builtin_inff();
    if (a.fin.C > 0) {  // every workgroup finalises the BatchNorm's backward from its accumulators (requested above)
        bnfin_backward<TH_THREADS>(a.fin, finr, cgr, cgr + 4 * Cgr_t + 4);
    } else {
        for (int e = tid; e < 4 * Cgr_t; e += TH_THREADS) {
            const int q = e / Cgr_t, c = e - q * Cgr_t;
            float v = q == 1 ? 1.f : 0.f;
            if (tgr.mode == TR_GRAD_BN) v = tgr.coef[(q == 0 ? 0 : 3 + q) * Cgr_t + c];
            cgr[e] = v;
        }
    }
    // per-slot constants: channel and LDS destination (a round's S and G are contiguous in global memory:
    // [sample][channel][pixel], and S keeps that layout in LDS)
    int sch[NS], gch[NG], gdst[NG];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int e = min(tid + 256 * j, nS - 1);
        const int sl = (int)fdiv((uint32_t)e, a.dS_CHW);
        sch[j] = (int)fdiv((uint32_t)(e - sl * nS1), a.dS_HW);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int e = min(tid + 256 * j, nG - 1);
        const int sl = (int)fdiv((uint32_t)e, a.dG_CHW), r = e - sl * nG1;
        const int c = (int)fdiv((uint32_t)r, a.dG_HW), pos = r - c * GHW;
        const int ih = (int)fdiv((uint32_t)pos, a.dG_W), iw = pos - ih * a.GW;
        gch[j] = c;
        gdst[j] = sl * imgG1 + c * HpWp + (ih + a.ph) * a.Wp + iw + a.pw;
    }
    // MFMA index sets: this wave's k-steps are wid, wid + 4, ...; pixel of a step for this lane: 4 * step + lk
    int pS[MID_KSW], pG[MID_KSW], pP[MID_KSW];
#pragma unroll
    for (int q = 0; q < MID_KSW; ++q) {
        const int p = 4 * (wid + 4 * q) + lk;  // pixel of the round: (sample, oh, ow)
        const int sl = (int)fdiv((uint32_t)p, a.dS_HW);
        const bool on = sl < a.SBR;
        const int pix = on ? p - sl * SHW : 0, sl0 = on ? sl : 0;
        const int oh = (int)fdiv((uint32_t)pix, a.dS_W), ow = pix - oh * a.SW;
        pS[q] = sl0 * nS1 + pix;
        pG[q] = sl0 * imgG1 + oh * a.sh * a.Wp + ow * a.sw;
        pP[q] = on ? p : 0x7fffffff;
    }
    int moff[MT], koffN[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) moff[mt] = min(mt * 16 + li, a.Sc - 1) * SHW;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = min(nt * 16 + li, NK - 1);
        const int c = (int)fdiv((uint32_t)n, a.d_KK), tap = n - c * KK;
        const int kh = (int)fdiv((uint32_t)tap, a.d_KW), kw = tap - kh * a.KW;
        koffN[nt] = c * HpWp + kh * a.Wp + kw;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    constexpr int NB = S_GRAD ? NS : NG;  // bias partials: one per gradient slot (its channel is fixed)
    float bacc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bacc[j] = 0.f;
    lds_barrier();
    IG_T();
    const float thr = cin[4 * Cin_t];

    // transform + store one round's registers into buffer b; the gradient values also feed the bias partials
    auto store = [&](int b, int rd, const float (&sv)[NS], const float (&sy)[S_GRAD ? NS : 1], const float (&gv)[NG],
                     const float (&gy)[S_GRAD ? 1 : NG]) {
        const int sbr = rd <= last ? min(a.SBR, a.B - rd * a.SBR) : 0;  // (0: nothing of this round counts)
        float so[NS], go[NG];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int c = sch[j];
            if (S_GRAD) {
                const float y = sy[j];
                so[j] = fmaf(cgr[Cgr_t + c], sv[j], fmaf(cgr[2 * Cgr_t + c], y - cgr[c], cgr[3 * Cgr_t + c])) * th_act_grad<ACT>(y, a.slope);
            } else {
                so[j] = fmaxf(fmaf(sv[j] - cin[c], cin[2 * Cin_t + c], cin[3 * Cin_t + c]), thr);
            }
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int c = gch[j];
            if (!S_GRAD) {
                const float y = gy[j];
                go[j] = fmaf(cgr[Cgr_t + c], gv[j], fmaf(cgr[2 * Cgr_t + c], y - cgr[c], cgr[3 * Cgr_t + c])) * th_act_grad<ACT>(y, a.slope);
            } else {
                go[j] = fmaxf(fmaf(gv[j] - cin[c], cin[2 * Cin_t + c], cin[3 * Cin_t + c]), thr);
            }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (tid + 256 * j < nS) lS[b][tid + 256 * j] = so[j];
            if (S_GRAD) bacc[j] += tid + 256 * j < sbr * nS1 ? so[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            if (tid + 256 * j < nG) lG[b][gdst[j]] = go[j];
            if (!S_GRAD) bacc[j] += tid + 256 * j < sbr * nG1 ? go[j] : 0.f;
        }
    };
    auto compute = [&](int b, int rd) {
        if (rd > last) return;
        const int P = min(a.SBR, a.B - rd * a.SBR) * SHW;  // pixels of this round
        const float* S = lS[b];
        const float* G = lG[b];
#pragma unroll
        for (int q = 0; q < MID_KSW; ++q) {
            if (4 * (wid + 4 * q) >= P) break;  // (wave-uniform)
            const float on = pP[q] < P ? 1.f : 0.f;
            float av[MT], bv[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = S[moff[mt] + pS[q]] * on;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = G[pG[q] + koffN[nt]];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
    };

    store(0, blockIdx.x, sA, sAy, gA, gAy);
    IG_T();
    load(blockIdx.x + 2 * g, sA, sAy, gA, gAy);
    for (int rd = blockIdx.x; rd <= last; rd += 2 * g) {
        lds_barrier();  // buffer 0 is complete, buffer 1 is free
        IG_T();
        compute(0, rd);
        IG_T();
        store(1, rd + g, sB, sBy, gB, gBy);
        load(rd + 3 * g, sB, sBy, gB, gBy);
        lds_barrier();
        IG_T();
        compute(1, rd + g);
        IG_T();
        store(0, rd + 2 * g, sA, sAy, gA, gAy);
        load(rd + 4 * g, sA, sAy, gA, gAy);
    }
    IG_T();

    // ---- cross-wave reduction (the staging buffers are dead), MT*NT tiles in passes of <= 5, and the partial row
    constexpr int TE = MT * NT * 256, RT = MT * NT < 5 ? MT * NT : 5;
    float* red = smem;  // [4][RT * 256] (the host sizes the dynamic LDS for it)
    float* row = a.part + (int64_t)blockIdx.x * (TE + 64);
#pragma unroll
    for (int t0 = 0; t0 < MT * NT; t0 += RT) {
        __syncthreads();
#pragma unroll
        for (int t = t0; t < t0 + RT && t < MT * NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wid * RT * 256 + ((t - t0) * 4 + r) * 64 + lane] = acc[t / NT][t % NT][r];
        __syncthreads();
        const int cnt = min(RT, MT * NT - t0) * 256;
        for (int i = tid; i < cnt; i += TH_THREADS)
            row[t0 * 256 + i] = (red[i] + red[RT * 256 + i]) + (red[2 * RT * 256 + i] + red[3 * RT * 256 + i]);
    }
    __syncthreads();
    // bias gradient: every slot's partial goes to its element's place in the round, then one thread per channel adds
    // its planes (fixed order)
    if (a.want_bias) {
        const int nB = S_GRAD ? nS : nG, nB1 = S_GRAD ? nS1 : nG1, HWb = S_GRAD ? SHW : GHW, Cb = S_GRAD ? a.Sc : a.Gc;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (tid + 256 * j < nB) red[tid + 256 * j] = bacc[j];
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            if (tid < Cb)
                for (int sl = 0; sl < a.SBR; ++sl)
                    for (int i = 0; i < HWb; ++i) t += red[sl * nB1 + tid * HWb + i];
            row[TE + tid] = t;
        }
    } else if (tid < 64) {
        row[TE + tid] = 0.f;
    }
    IG_T();
#ifdef IG_TIMERS
    if (threadIdx.x == 0 && blockIdx.x == 1) printf("mid_wgrad<%d,%d,%d> grid %d rounds %d SBR %d: issue | setup | store0 | (barrier compute barrier compute)* | loop end | reduce ...", MT, NT, (int)S_GRAD, g, nrounds, a.SBR);
#endif
    IG_TPRINT("");
}

// mid_conv_k<GRADIN, EPBN, ACT>: plain (non-transposed) gather of a mid-size layer: Conv forward (conv2) and ConvT input
// gradient (convT2).  One sample per round; the input goes through a workgroup-shared zero-haloed LDS image
// (double-buffered, two rounds of loads in flight); wave w owns the 16-pixel tile w of the sample's <= 64 output pixels
// and keeps its 18 k-steps of weights (A operand) and im2col offsets in registers, so a tile is 18 x (ds_read +
// v_mfma_f32_16x16x4_f32).  Results leave from the accumulators: D row = channel lk*4 + r, column = pixel li.
//   GRADIN: the input is a gradient: (ga*g + gb*(y - mean) + gc) * act'(y) on load (ACT = that layer's activation)
//   EPBN  : backward through the producer's virtual BatchNorm (+ReLU): mask, and its two sums per channel (slab rows
//           {sum g, sum g*xhat}); else bias + activation (ACT) and, with a slab, forward statistics
//           {pivot = bias, sum(y-p), sum((y-p)^2), n}.  One slab row per workgroup.
// Uses IgArgs (igemm.h) for the geometry, tensors and epilogue description; its chunk/LDS/class fields are ignored.
#define MIDC_KS 18  // Cin * KH * KW <= 72
#define MIDC_NE 7   // Cin * H * W <= 1792
template <bool GRADIN, bool EPBN, int ACT>
__global__ __launch_bounds__(TH_THREADS) void mid_conv_k(IgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const IgGeom& g = a.g;
    IG_TDECL;
    IG_T();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int HW = g.H * g.W, CHW = g.Cin * HW, HpWp = g.Hp * g.Wp, OHW = g.OH * g.OW, KK = g.KH * g.KW, K = g.Cin * KK;
    const int imgN = g.Cin * HpWp;
    float* img[2] = {smem, smem + imgN};
    float* cin = smem + 2 * imgN;                 // input coefficients [4 * Cin + 4]
    float* cep = cin + 4 * g.Cin + 4;             // EPBN: producer BatchNorm [4 * N + 4] (mean, rstd, scale, beta, thr)
    float* red = cep + 4 * g.N + 4;               // [4][16][2] + [4]
    const int G = gridDim.x, last = a.B - 1;

    float vA[MIDC_NE], vB[MIDC_NE], yA[GRADIN ? MIDC_NE : 1], yB[GRADIN ? MIDC_NE : 1];
    float eA[EPBN ? 4 : 1], eB[EPBN ? 4 : 1];  // EPBN: the producer's outputs at this lane's four (channel, pixel) results
    const int p_out = wid * 16 + li;           // this lane's output pixel
    const bool p_ok = p_out < OHW;
    auto load = [&](int s, float (&v)[MIDC_NE], float (&y)[GRADIN ? MIDC_NE : 1], float (&e)[EPBN ? 4 : 1]) {
        s = min(s, last);
        // (opaque thread index: otherwise the optimiser hoists every slot's 64-bit address out of the round loop,
        // once per inlined copy of this lambda, into ~100 registers)
        const int t0 = (int)opaque_tid();
        const float* src = a.in.src + (int64_t)s * CHW;
        const float* src2 = GRADIN ? a.in.src2 + (int64_t)s * CHW : nullptr;
#pragma unroll
        for (int j = 0; j < MIDC_NE; ++j) {
            const int i = min(t0 + 256 * j, CHW - 1);
            v[j] = src[i];
            if (GRADIN) y[j] = src2[i];
        }
        if (EPBN) {
            const float* py = a.ep_y + (int64_t)s * g.N * OHW + (p_ok ? p_out : 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = py[min(lk * 4 + r, g.N - 1) * OHW];
        }
    };
    load(blockIdx.x, vA, yA, eA);
    load(blockIdx.x + G, vB, yB, eB);
    IG_T();

    // ---- once per workgroup
    for (int e = tid; e < 2 * imgN; e += TH_THREADS) smem[e] = 0.f;  // the halo stays zero
    if (GRADIN) {
        for (int e = tid; e < 4 * g.Cin; e += TH_THREADS) {
            const int q = e / g.Cin, c = e - q * g.Cin;
            float v = q == 1 ? 1.f : 0.f;
            if (a.in.mode == TR_GRAD_BN) v = a.in.coef[(q == 0 ? 0 : 3 + q) * g.Cin + c];
            cin[e] = v;
        }
    } else {
        for (int e = tid; e < 4 * g.Cin; e += TH_THREADS) {
            float v = (e >= 2 * g.Cin && e < 3 * g.Cin) ? 1.f : 0.f;
            if (a.in.mode == TR_BN) v = a.in.coef[e];
            cin[e] = v;
        }
        if (tid == 0) cin[4 * g.Cin] = (a.in.mode == TR_BN && a.in.relu) ? 0.f : -__builtin_inff();
    }
    if (EPBN) {
        for (int e = tid; e < 4 * g.N; e += TH_THREADS) cep[e] = a.ep_coef[e];
        if (tid == 0) cep[4 * g.N] = a.ep_relu ? 0.f : -__builtin_inff();
    }
    IG_T();
    int sch[MIDC_NE], sdst[MIDC_NE];
#pragma unroll
    for (int j = 0; j < MIDC_NE; ++j) {
        const int e = min(tid + 256 * j, CHW - 1);
        const int c = (int)fdiv((uint32_t)e, g.d_HW), pos = e - c * HW;
        const int ih = (int)fdiv((uint32_t)pos, g.d_W), iw = pos - ih * g.W;
        sch[j] = c;
        sdst[j] = c * HpWp + (ih + g.halo_h) * g.Wp + iw + g.halo_w;
    }
    IG_T();
    float wreg[MIDC_KS];
    int koff[MIDC_KS];
    // (all 18 weight loads are issued before any of them is used: with the mask applied in the same loop the
    // compiler waited for every load in turn, 18 x 270 ns of a 6 us set-up)
#pragma unroll
    for (int s = 0; s < MIDC_KS; ++s) {
        const int k = min(4 * s + lk, K - 1);  // (cls[0] of a plain gather: ntaps = KH * KW, nkw = KW)
        const int ci = (int)fdiv((uint32_t)k, a.cls[0].d_ntaps), tap = k - ci * KK;
        const int kh = (int)fdiv((uint32_t)tap, a.cls[0].d_nkw), kw = tap - kh * g.KW;
        koff[s] = ci * HpWp + kh * g.Wp + kw;
        wreg[s] = a.w[ci * g.w_cstride + min(li, g.N - 1) * g.w_nstride + tap];
    }
#pragma unroll
    for (int s = 0; s < MIDC_KS; ++s) wreg[s] = (4 * s + lk < K && li < g.N) ? wreg[s] : 0.f;
    IG_T();
    const int oh = (int)fdiv((uint32_t)(p_ok ? p_out : 0), a.cls[0].d_AW), ow = (p_ok ? p_out : 0) - oh * g.OW;
    const int base = oh * g.si_h * g.Wp + ow * g.si_w;
    float br[4], m_[4], rs_[4], sc_[4], be_[4], s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ch = min(lk * 4 + r, g.N - 1);
        br[r] = (!EPBN && a.bias != nullptr) ? a.bias[ch] : 0.f;
        s1[r] = s2[r] = 0.f;
        m_[r] = rs_[r] = sc_[r] = be_[r] = 0.f;
    }
    lds_barrier();
    if (EPBN) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = min(lk * 4 + r, g.N - 1);
            m_[r] = cep[ch]; rs_[r] = cep[g.N + ch]; sc_[r] = cep[2 * g.N + ch]; be_[r] = cep[3 * g.N + ch];
        }
    }
    const float thr_in = GRADIN ? 0.f : cin[4 * g.Cin];
    const float thr_ep = EPBN ? cep[4 * g.N] : 0.f;
    float cnt = 0.f;

    auto store = [&](float* t, const float (&v)[MIDC_NE], const float (&y)[GRADIN ? MIDC_NE : 1]) {
        float o[MIDC_NE];
#pragma unroll
        for (int j = 0; j < MIDC_NE; ++j) {
            const int c = sch[j];
            if (GRADIN)
                o[j] = fmaf(cin[g.Cin + c], v[j], fmaf(cin[2 * g.Cin + c], y[j] - cin[c], cin[3 * g.Cin + c])) * th_act_grad<ACT>(y[j], a.in.slope);
            else
                o[j] = fmaxf(fmaf(v[j] - cin[c], cin[2 * g.Cin + c], cin[3 * g.Cin + c]), thr_in);
        }
#pragma unroll
        for (int j = 0; j < MIDC_NE; ++j)
            if (tid + 256 * j < CHW) t[sdst[j]] = o[j];
    };
    auto compute = [&](const float* t, int s, const float (&e)[EPBN ? 4 : 1]) {
        if (s > last) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* p = t + base;
#pragma unroll
        for (int q = 0; q < MIDC_KS; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[q], p[koff[q]], acc, 0, 0, 0);
        float* o = a.out + (int64_t)s * g.N * OHW + (p_ok ? p_out : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = p_ok && lk * 4 + r < g.N;
            if (EPBN) {
                const float xc = e[r] - m_[r];
                float d = fmaf(xc, sc_[r], be_[r]) > thr_ep ? acc[r] : 0.f;
                d = ok ? d : 0.f;
                s1[r] += d;
                s2[r] = fmaf(d, xc * rs_[r], s2[r]);
                if (ok) o[(lk * 4 + r) * OHW] = d;
            } else {
                // (GRADIN: ACT is the activation of the layer whose gradient comes in; the result has none)
                const float v = GRADIN ? acc[r] + br[r] : th_act<ACT>(acc[r] + br[r], a.slope);
                if (ok) o[(lk * 4 + r) * OHW] = v;
                const float d = ok ? v - br[r] : 0.f;  // statistics pivot: the channel's bias
                s1[r] += d;
                s2[r] = fmaf(d, d, s2[r]);
            }
        }
        cnt += (float)OHW;
    };

    IG_T();
    store(img[0], vA, yA);
    float eC[EPBN ? 4 : 1];  // (the epilogue values of the sample being computed survive the reload of their set)
#pragma unroll
    for (int r = 0; r < (EPBN ? 4 : 1); ++r) eC[r] = eA[r];
    load(blockIdx.x + 2 * G, vA, yA, eA);
    for (int s = blockIdx.x; s <= last; s += 2 * G) {
        lds_barrier();  // image 0 is complete, image 1 is free
        IG_T();
        compute(img[0], s, eC);
        IG_T();
        store(img[1], vB, yB);
#pragma unroll
        for (int r = 0; r < (EPBN ? 4 : 1); ++r) eC[r] = eB[r];
        load(s + 3 * G, vB, yB, eB);
        IG_T();
        lds_barrier();
        IG_T();
        compute(img[1], s + G, eC);
        IG_T();
        store(img[0], vA, yA);
#pragma unroll
        for (int r = 0; r < (EPBN ? 4 : 1); ++r) eC[r] = eA[r];
        load(s + 4 * G, vA, yA, eA);
    }

    IG_T();
    IG_TPRINT("mid_conv: setup | [barrier] compute store+load [barrier] compute (store+load) ...");
    if (a.slab == nullptr) return;  // (uniform)
    // ---- per-channel sums: over the 16 pixel lanes of a channel row (DPP), then over the four waves
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s1[r] = dpp_add<0x111, 0xf>(s1[r]); s1[r] = dpp_add<0x112, 0xf>(s1[r]);
        s1[r] = dpp_add<0x114, 0xf>(s1[r]); s1[r] = dpp_add<0x118, 0xf>(s1[r]);
        s2[r] = dpp_add<0x111, 0xf>(s2[r]); s2[r] = dpp_add<0x112, 0xf>(s2[r]);
        s2[r] = dpp_add<0x114, 0xf>(s2[r]); s2[r] = dpp_add<0x118, 0xf>(s2[r]);
        if (li == 15) {  // lane 15 of a row holds the row's sum
            red[(wid * 16 + lk * 4 + r) * 2] = s1[r];
            red[(wid * 16 + lk * 4 + r) * 2 + 1] = s2[r];
        }
    }
    if (lane == 0) red[128 + wid] = cnt;
    __syncthreads();
    if (tid < g.N) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            t1 += red[(w * 16 + tid) * 2];
            t2 += red[(w * 16 + tid) * 2 + 1];
        }
        float* sl = a.slab + ((int64_t)blockIdx.x * g.N + tid) * 4;
        if (EPBN) {
            sl[0] = t1; sl[1] = t2; sl[2] = 0.f; sl[3] = 0.f;
            if (a.acc_save != nullptr) bnacc_add_bwd(a.acc_save, g.N, tid, t1, t2);
        } else {
            const float bp = a.bias != nullptr ? a.bias[tid] : 0.f, cn = red[128];  // every wave counted the workgroup's samples
            sl[0] = bp;
            sl[1] = t1; sl[2] = t2; sl[3] = cn;
            if (a.acc_save != nullptr) {
                const float dd = t1 / cn, pv = a.acc_pivot[tid];
                bnacc_add_fwd(a.acc_save, g.N, tid, cn, bp + dd, fmaxf(t2 - t1 * dd, 0.f), pv);
                if (blockIdx.x == 0) a.acc_save[7 * g.N + tid] = pv;
            }
        }
    }
}
