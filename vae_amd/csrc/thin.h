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
    load_own(blockIdx.x, vA);
    load_own(min((int)blockIdx.x + g, last), vB);
    for (int e = tid; e < 2 * tile_n; e += TH_THREADS) smem[e] = 0.f;
    th_in_coef<C>(a.in, coef);
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
    }
}
