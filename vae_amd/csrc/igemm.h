// igemm.h - LDS-staged implicit-GEMM convolution kernels on fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// One workgroup owns a chunk of SB consecutive samples.  It stages the chunk's input
// activations ONCE into LDS (zero halo, with the producer's BatchNorm+ReLU or this layer's
// gradient transform applied on the way in), builds the im2col offset table and the
// [k][n] weight image in LDS, and then every 16-pixel x 16-channel output tile is a chain of
// MFMAs whose A operand is gathered from the staged tile (pixel base + tap offset) and whose
// B operand is the weight image.  Results go through an LDS output tile so that the global
// store of the chunk is one contiguous, coalesced stream, and the per-channel sums that
// BatchNorm needs (forward: sum, sum of squares; backward: sum g, sum g*xhat) are taken from
// that tile in the same kernel ("virtual" BatchNorm: its output is never materialised).
//
//   igemm_gather_k : Conv forward, ConvT forward (per output-parity class), and both input
//                    gradients (Conv dgrad = transposed gather, ConvT dgrad = conv gather).
//   igemm_wgrad_k  : weight gradient (K = pixels), bias gradient taken while staging.
#pragma once
#include <type_traits>
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
// bf16 compute mode (vae_set_compute_dtype): the MFMA operands are rounded to bf16 (round to nearest even,
// v_cvt_pk_bf16_f32) when they leave LDS, products accumulate in fp32 (v_mfma_f32_16x16x16_bf16: 16 k-values per
// instruction at 8x the fp32 matrix rate per k-value); everything stored in memory stays fp32.  Operand layout: lane
// l supplies row/column l & 15 and the FOUR CONSECUTIVE k-values (l >> 4) * 4 + 0..3 of a 16-wide k block (the fp32
// instruction takes one k-value per lane: (l >> 4) of a 4-wide block); the D layout is the same.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ s16x4 bf16_pack4(float a, float b, float c, float d) {
    bf16x4 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    return __builtin_bit_cast(s16x4, h);
}
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)

enum { TR_PLAIN = 0, TR_BN = 1, TR_GRAD_BN = 2, TR_GRAD_ACT = 3 };
enum { EP_PLAIN = 0, EP_BNGRAD = 1 };

// How a [B][C][HW] tensor is read into LDS.
//   TR_PLAIN    v = src[e]
//   TR_BN       v = (src[e]-mean[c])*scale[c] + beta[c], then ReLU if relu   (coef = BN save block)
//   TR_GRAD_BN  v = (a[c]*src[e] + b[c]*(src2[e]-mean[c]) + cc[c]) * act'(src2[e])   (src = g, src2 = y)
//   TR_GRAD_ACT v = src[e] * act'(src2[e])
struct TensorIn {
    const float* src;
    const float* src2;
    const float* coef;  // BN save block: mean, rstd, scale, beta, a, b, c, spare (each C floats)
    int mode, relu, act;
    float slope;
};

// `coef` is the LDS copy of t.coef (8*C floats) made once per workgroup.  v = src[e], y = src2[e]
// (y is only read by the gradient modes).
__device__ __forceinline__ float tensor_transform(const TensorIn& t, const float* coef, float v, float y, int c, int C) {
    if (t.mode == TR_BN) {
        v = fmaf(v - coef[c], coef[2 * C + c], coef[3 * C + c]);  // (y - mean) * scale + beta
        if (t.relu) v = v > 0.f ? v : 0.f;
    } else if (t.mode == TR_GRAD_BN) {
        v = fmaf(coef[4 * C + c], v, fmaf(coef[5 * C + c], y - coef[c], coef[6 * C + c]));
        v *= act_grad_from_out(y, t.act, t.slope);
    } else if (t.mode == TR_GRAD_ACT) {
        v *= act_grad_from_out(y, t.act, t.slope);
    }
    return v;
}
__device__ __forceinline__ float tensor_in_value(const TensorIn& t, const float* coef, int64_t e, int c, int C) {
    const float v = t.src[e];
    const float y = t.mode >= TR_GRAD_BN ? t.src2[e] : 0.f;
    return tensor_transform(t, coef, v, y, c, C);
}

struct FastDiv {  // exact n / d for n * d < 2^32
    uint32_t mul, d;
};
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return f.d == 1 ? n : __umulhi(n, f.mul); }

// ---- a virtual BatchNorm finalised inside its first consumer's prologue ---------------------------------------
// A finalize launch of its own (bnv_finalize_*_k) costs 4.6-6.5 us plus a kernel boundary on KB-sized data, six times
// per MNIST step.  Merging the producers' per-workgroup rows in every consumer workgroup's prologue was measured and
// lost (512 rows x 16 channels per workgroup: +4..13 us per consumer, 0.521 -> 0.545 ms/step).  What is built: besides
// its row, every producer workgroup adds its per-channel sums to FIXED-POINT 64-bit accumulators with integer atomics
// (integer addition is associative: the totals are bit-identical from run to run whatever the arrival order, unlike
// float atomics), sharded sixteen ways by blockIdx.x & 15, every word in a 64-byte line of its own (atomics execute at
// the memory side, one line at a time: with the three words of a channel in one line and eight shards the producers'
// tails grew by 2-4 us).  The first consumer kernel then reads 16 x 3 words per channel in its prologue (requested at
// the top of the kernel, used after its set-up; every workgroup the same words, so all hold bit-identical coefficients) and workgroup 0 stores the save block, the running
// statistics (forward) or dgamma / dbeta (backward) for the kernels that follow.  The accumulators live right behind
// the BatchNorm's save block (8*C floats): [fwd | bwd][quantity 3][shard 16][C][line of 8 words]; the host zeroes them at the
// start of every training-mode forward pass (one memset for all BatchNorms of a program).
//   forward : word 0 = n, 1 = sum n_w*(mean_w - p) * 2^24, 2 = sum (M2_w + n_w*(mean_w - p)^2) * 2^24, with the pivot
//             p = the BatchNorm's running mean before this step (producer workgroup 0 leaves it in save[7*C + c])
//   backward: word 0 = sum g * 2^32, 1 = sum g*xhat * 2^32
// Ranges: |sum| < 2^63 / scale = 5.5e11 (forward: n * (var + (mean - p)^2)) and 2.1e9 (backward sums).
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    const float tot = n + nb;
    if (tot > 0.f) {
        const float delta = mb - mean;
        const float f = nb / tot;
        mean = fmaf(delta, f, mean);
        m2 += m2b + delta * delta * n * f;
    }
    n = tot;
}
#define BNACC_SHARDS 16
#define BNACC_NW 3     // quantities per (shard, channel)
#define BNACC_LINE 8   // 8-byte words per 64-byte line: every (quantity, shard, channel) word has a line of its own
#define BNACC_DIR_WORDS(C) (BNACC_NW * BNACC_SHARDS * (C) * BNACC_LINE)
#define BNACC_FLOATS(C) (2 * BNACC_DIR_WORDS(C) * 2)  // forward + backward accumulators, in floats
#define BNACC_FWD_SCALE 16777216.0    // 2^24
#define BNACC_BWD_SCALE 4294967296.0  // 2^32
typedef unsigned long long bnacc_t;
__device__ __forceinline__ bnacc_t* bnacc_fwd(const float* save, int C) {
    return reinterpret_cast<bnacc_t*>(const_cast<float*>(save) + 8 * C);
}
__device__ __forceinline__ bnacc_t* bnacc_bwd(const float* save, int C) { return bnacc_fwd(save, C) + BNACC_DIR_WORDS(C); }
__device__ __forceinline__ int bnacc_idx(int C, int word, int shard, int c) { return ((word * BNACC_SHARDS + shard) * C + c) * BNACC_LINE; }
// one producer workgroup's (n, mean, M2) of channel c (called by one lane per channel)
__device__ __forceinline__ void bnacc_add_fwd(const float* save, int C, int c, float n, float mean, float m2, float pivot) {
    bnacc_t* p = bnacc_fwd(save, C);
    const int sh = blockIdx.x & (BNACC_SHARDS - 1);
    const double dm = (double)mean - (double)pivot, dn = (double)n;
    atomicAdd(p + bnacc_idx(C, 0, sh, c), (bnacc_t)(long long)n);
    atomicAdd(p + bnacc_idx(C, 1, sh, c), (bnacc_t)__double2ll_rn(dn * dm * BNACC_FWD_SCALE));
    atomicAdd(p + bnacc_idx(C, 2, sh, c), (bnacc_t)__double2ll_rn(((double)m2 + dn * dm * dm) * BNACC_FWD_SCALE));
}
__device__ __forceinline__ void bnacc_add_bwd(const float* save, int C, int c, float s1, float s2) {
    bnacc_t* p = bnacc_bwd(save, C);
    const int sh = blockIdx.x & (BNACC_SHARDS - 1);
    atomicAdd(p + bnacc_idx(C, 0, sh, c), (bnacc_t)__double2ll_rn((double)s1 * BNACC_BWD_SCALE));
    atomicAdd(p + bnacc_idx(C, 1, sh, c), (bnacc_t)__double2ll_rn((double)s2 * BNACC_BWD_SCALE));
}
struct BnFin {
    const float* slab;  // the producers' rows [nrows][C][4]: only the host's fallback (bn_flush_pending) reads them
    int nrows, C, training;
    const float* gamma;
    const float* beta;
    float* save;     // [8*C]: mean, rstd, scale, beta, a, b, c, pivot; the accumulators follow
    float* rmean;    // forward: running statistics
    float* rvar;
    float* dgamma;   // backward
    float* dbeta;
    float eps, momentum, n_local;
};
#define BNFIN_MAXC 16
#define BNFIN_SCR_FLOATS (BNACC_NW * BNACC_SHARDS * BNFIN_MAXC * 2)  // LDS scratch of a finalize prologue (8-byte aligned)
// The consumer's prologue in two parts: bnfin_issue at the very top of the kernel (thread t < 16*C requests the three
// words of (shard t / C, channel t % C); nothing waits on them), bnfin_forward / bnfin_backward after the kernel's
// own set-up, where their latency has passed.
struct BnFinRegs {
    long long a0, a1, a2;
};
template <bool BWD>
__device__ __forceinline__ void bnfin_issue(const BnFin& f, BnFinRegs& r) {
    r.a0 = r.a1 = r.a2 = 0;
    const int C = f.C, tid = threadIdx.x;
    if (C > 0 && f.training && tid < BNACC_SHARDS * C) {
        const long long* p = reinterpret_cast<const long long*>(BWD ? bnacc_bwd(f.save, C) : bnacc_fwd(f.save, C));
        const int sh = tid / C, c = tid - sh * C;
        r.a0 = p[bnacc_idx(C, 0, sh, c)];
        r.a1 = p[bnacc_idx(C, 1, sh, c)];
        if (!BWD) r.a2 = p[bnacc_idx(C, 2, sh, c)];
    }
}
// coef (LDS): [0,C) mean, [C,2C) rstd, [2C,3C) scale, [3C,4C) beta.  scr (LDS): BNFIN_SCR_FLOATS.  All T >= 16*C
// threads call; ends with a workgroup barrier.
template <int T>
__device__ __forceinline__ void bnfin_forward(const BnFin& f, const BnFinRegs& r, float* coef, float* scr) {
    static_assert(T >= BNACC_SHARDS * BNFIN_MAXC, "one thread per (shard, channel)");
    const int C = f.C, tid = threadIdx.x;
    long long* w = reinterpret_cast<long long*>(scr);
    if (f.training) {
        if (tid < BNACC_SHARDS * C) { w[3 * tid] = r.a0; w[3 * tid + 1] = r.a1; w[3 * tid + 2] = r.a2; }
        __syncthreads();
        if (tid < C) {
            long long s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < BNACC_SHARDS; ++k) { s0 += w[3 * (k * C + tid)]; s1 += w[3 * (k * C + tid) + 1]; s2 += w[3 * (k * C + tid) + 2]; }
            const double n = (double)s0, d1 = (double)s1 / BNACC_FWD_SCALE, d2 = (double)s2 / BNACC_FWD_SCALE;
            const double dmean = d1 / n;
            const float mean = (float)((double)f.save[7 * C + tid] + dmean);
            const float var = (float)(fmax(d2 - d1 * dmean, 0.0) / n);
            const float rstd = rsqrtf(var + f.eps), sc = f.gamma[tid] * rstd, be = f.beta[tid];
            coef[tid] = mean; coef[C + tid] = rstd; coef[2 * C + tid] = sc; coef[3 * C + tid] = be;
            if (blockIdx.x == 0) {
                f.save[tid] = mean; f.save[C + tid] = rstd; f.save[2 * C + tid] = sc; f.save[3 * C + tid] = be;
                const float nf = (float)n, unbiased = nf > 1.f ? var * nf / (nf - 1.f) : var;
                f.rmean[tid] = (1.f - f.momentum) * f.rmean[tid] + f.momentum * mean;
                f.rvar[tid] = (1.f - f.momentum) * f.rvar[tid] + f.momentum * unbiased;
            }
        }
    } else if (tid < C) {  // eval: running statistics
        const float mean = f.rmean[tid], rstd = rsqrtf(f.rvar[tid] + f.eps), sc = f.gamma[tid] * rstd, be = f.beta[tid];
        coef[tid] = mean; coef[C + tid] = rstd; coef[2 * C + tid] = sc; coef[3 * C + tid] = be;
        if (blockIdx.x == 0) { f.save[tid] = mean; f.save[C + tid] = rstd; f.save[2 * C + tid] = sc; f.save[3 * C + tid] = be; }
    }
    __syncthreads();
}
// cgr (LDS): [0,C) mean, [C,2C) a, [2C,3C) b, [3C,4C) c of the gradient transform g' = a*g + b*(y - mean) + c.
template <int T>
__device__ __forceinline__ void bnfin_backward(const BnFin& f, const BnFinRegs& r, float* cgr, float* scr) {
    static_assert(T >= BNACC_SHARDS * BNFIN_MAXC, "one thread per (shard, channel)");
    const int C = f.C, tid = threadIdx.x;
    long long* w = reinterpret_cast<long long*>(scr);
    if (tid < BNACC_SHARDS * C) { w[2 * tid] = r.a0; w[2 * tid + 1] = r.a1; }
    __syncthreads();
    if (tid < C) {
        long long t1 = 0, t2 = 0;
#pragma unroll
        for (int k = 0; k < BNACC_SHARDS; ++k) { t1 += w[2 * (k * C + tid)]; t2 += w[2 * (k * C + tid) + 1]; }
        const float s1 = (float)((double)t1 / BNACC_BWD_SCALE), s2 = (float)((double)t2 / BNACC_BWD_SCALE);
        const float m1 = s1 / f.n_local, m2 = s2 / f.n_local;
        const float mean = f.save[tid], rstd = f.save[C + tid], sc = f.save[2 * C + tid];
        const float ga = sc, gb = -sc * m2 * rstd, gc = -sc * m1;
        cgr[tid] = mean; cgr[C + tid] = ga; cgr[2 * C + tid] = gb; cgr[3 * C + tid] = gc;
        if (blockIdx.x == 0) {
            f.dbeta[tid] = s1;
            f.dgamma[tid] = s2;
            f.save[4 * C + tid] = ga; f.save[5 * C + tid] = gb; f.save[6 * C + tid] = gc;
        }
    }
    __syncthreads();
}

struct IgGeom {
    int Cin, H, W;      // gathered tensor, per sample
    int N, OH, OW;      // produced tensor
    int KH, KW;
    int si_h, si_w;     // input step per class pixel  (conv: stride, transposed: 1)
    int so_h, so_w;     // output step per class pixel (conv: 1, transposed: stride)
    int ph, pw;
    int transposed;
    int Hp, Wp, halo_h, halo_w;  // LDS image of one input channel: Hp x Wp with zero halo
    int w_cstride, w_nstride;    // weight element (ci, n, kh, kw) at ci*w_cstride + n*w_nstride + kh*KW + kw
    int Np;                      // N rounded up to 16
    FastDiv d_HW, d_W, d_CHW;    // for staging decode
};

// One output-parity class (transposed gathers have so_h*so_w of them, plain convs one): which taps
// contribute and where its tap table / weight image start in LDS.  Filled on the host.
struct IgClass {
    int rh, rw, AH, AW, kh0, kw0, nkh, nkw, ntaps, Kc, KcPad, toff, woff;
    int poff;  // start of the class in the pixel table (filled at launch: SB * pixels of the earlier classes)
    int w0, nw;  // the waves [w0, w0 + nw) of a workgroup that compute this class's MFMA tiles (filled at launch)
    FastDiv d_ntaps, d_nkw, d_AHW, d_AW;
};

struct IgArgs {
    IgGeom g;
    TensorIn in;
    const float* w;
    const float* bias;
    float* out;
    int B, SB;  // SB consecutive samples form one chunk; workgroups loop over chunks
    int act;
    float slope;
    // epilogue
    int ep_mode;              // EP_PLAIN / EP_BNGRAD
    const float* ep_y;        // EP_BNGRAD: producer output y_prev (same layout as out)
    const float* ep_coef;     // EP_BNGRAD: producer BN save block
    int ep_relu;
    float* slab;              // per-block per-channel {pivot, s1, s2, count} (forward statistics) or
                              // {s1, s2, -, -} (backward sums), or nullptr
    // LDS carve (floats)
    int lds_in, lds_w, lds_out, lds_tab, lds_ptab;
    int ncls, nchunks;
    int pre;  // 1: chunks fit the register prefetch (<= IG_PRE*IG_THREADS input elements)
    FastDiv d_N, d_ntile;  // by N and by Np/16
    IgClass cls[4];
    BnFin fin;  // forward, in.mode == TR_BN: the producer BatchNorm is finalised in this kernel's prologue (C > 0)
    // the sums that go to `slab` are also added to the fixed-point accumulators behind this save block (nullptr: no):
    // forward statistics of the BatchNorm that follows (pivot: its running mean), or EP_BNGRAD's backward sums
    float* acc_save;
    const float* acc_pivot;
};

#ifdef IG_TIMERS  // diagnostic build only: phase stamps of workgroup 1 (wall clock, 10 ns ticks)
#define IG_TDECL long long tm_[24]; int tn_ = 0; const long long ck0_ = clock64()
#if IG_TIMERS >= 2  // wave 0's own timeline: no barrier and no wait in the stamp
#define IG_T() do { if (threadIdx.x == 0 && blockIdx.x == 1 && tn_ < 24) tm_[tn_++] = wall_clock64(); } while (0)
#else
#define IG_T() do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 1 && tn_ < 24) tm_[tn_++] = wall_clock64(); } while (0)
#endif
#define IG_TPRINT(name) do { if (threadIdx.x == 0 && blockIdx.x == 1) { printf("%s", name); for (int q_ = 1; q_ < tn_; ++q_) printf(" %lld", (tm_[q_] - tm_[q_-1]) * 10); printf(" ns; shader clock %.0f MHz\n", (double)(clock64() - ck0_) / ((double)(wall_clock64() - tm_[0]) * 10e-3)); } } while (0)
#else
#define IG_TDECL
#define IG_T()
#define IG_TPRINT(name)
#endif

#define IG_THREADS 512
#define IG_WAVES (IG_THREADS / 64)
#define IG_UNROLL 8
#define IG_PRE 16     // gather: prefetch registers per lane
#define WG_PRE 8      // wgrad: per tensor ...
// ... and 16 for the instances with 5-10 accumulator tiles: they run one workgroup per CU with a 256-register budget
// and were held to 2-sample chunks (8 rounds per workgroup, each paying the staging latency) by 8 registers
#define WG_PRE_OF(T) ((T) > WG_SMALL_T && (T) <= 10 ? 16 : WG_PRE)

// Stage [sb][C][H][W] (global, contiguous) into the padded LDS image, applying the transform.
// IG_UNROLL independent elements per lane and iteration: all their global loads are in flight
// together before the first LDS store (the kernels are latency-bound at these chunk sizes).
__device__ __forceinline__ void ig_stage(const TensorIn& t, const float* coef, float* lds, int64_t gbase, int sb,
                                         int C, int HW, int W, int Hp, int Wp, int halo_h, int halo_w,
                                         FastDiv d_CHW, FastDiv d_HW, FastDiv d_W) {
    const uint32_t total = (uint32_t)sb * C * HW;
    const int chw_p = C * Hp * Wp;
    for (uint32_t base = threadIdx.x; base < total; base += IG_THREADS * IG_UNROLL) {
        float v[IG_UNROLL];
        int dst[IG_UNROLL];
#pragma unroll
        for (int j = 0; j < IG_UNROLL; ++j) {
            const uint32_t e = base + j * IG_THREADS;
            dst[j] = -1;
            if (e < total) {
                const uint32_t s = fdiv(e, d_CHW);
                const uint32_t r = e - s * (uint32_t)(C * HW);
                const uint32_t c = fdiv(r, d_HW);
                const uint32_t pos = r - c * (uint32_t)HW;
                const uint32_t ih = fdiv(pos, d_W);
                const uint32_t iw = pos - ih * (uint32_t)W;
                v[j] = tensor_in_value(t, coef, gbase + e, (int)c, C);
                dst[j] = s * chw_p + c * Hp * Wp + (ih + halo_h) * Wp + iw + halo_w;
            }
        }
#pragma unroll
        for (int j = 0; j < IG_UNROLL; ++j)
            if (dst[j] >= 0) lds[dst[j]] = v[j];
    }
}

// Register prefetch of the NEXT chunk: pre_load only issues the global loads (nothing waits on them until
// pre_store, one chunk of compute later); pre_store transforms and writes the padded LDS image.  Where
// element (lane, j) lands does not depend on the chunk, so it is decoded once per workgroup into
// meta[j] = lds offset | channel << 16.  A tensor chunk holds <= PRE * IG_THREADS elements (host-checked).
// HASY: the tensor is read in a gradient mode (needs src2).
template <int PRE, bool HASY>
struct PreRegs {
    float v[PRE], y[HASY ? PRE : 1];
    uint32_t meta[PRE];
};
// keeps the compiler from hoisting per-slot address arithmetic out of the chunk loop into registers
__device__ __forceinline__ uint32_t opaque_tid() {
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
template <int PRE, bool HASY>
__device__ __forceinline__ void pre_decode(PreRegs<PRE, HASY>& p, uint32_t total, int C, int HW, int W, int Hp, int Wp,
                                           int halo_h, int halo_w, FastDiv d_CHW, FastDiv d_HW, FastDiv d_W) {
    const int chw_p = C * Hp * Wp;
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
        const uint32_t e = threadIdx.x + j * IG_THREADS;
        p.meta[j] = 0;
        if (e < total) {
            const uint32_t s = fdiv(e, d_CHW);
            const uint32_t r = e - s * (uint32_t)(C * HW);
            const uint32_t c = fdiv(r, d_HW);
            const uint32_t pos = r - c * (uint32_t)HW;
            const uint32_t ih = fdiv(pos, d_W);
            const uint32_t iw = pos - ih * (uint32_t)W;
            p.meta[j] = (s * chw_p + c * Hp * Wp + (ih + halo_h) * Wp + iw + halo_w) | (c << 16);
        }
    }
}
template <int PRE, bool HASY>
__device__ __forceinline__ void pre_load(const TensorIn& t, int64_t gbase, uint32_t total, PreRegs<PRE, HASY>& p) {
    const uint32_t t0 = opaque_tid();
    const float* src = t.src + gbase;
    const float* src2 = t.src2 != nullptr ? t.src2 + gbase : nullptr;
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
        const uint32_t e = t0 + j * IG_THREADS;
        if (e < total) {
            p.v[j] = src[e];
            if (HASY && t.mode >= TR_GRAD_BN) p.y[j] = src2[e];  // (src2 is null in the plain modes)
        }
    }
}
template <int PRE, bool HASY>
__device__ __forceinline__ void pre_store(const TensorIn& t, const float* coef, float* lds, uint32_t total, int C,
                                          const PreRegs<PRE, HASY>& p) {
    const uint32_t t0 = opaque_tid();
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
        const uint32_t e = t0 + j * IG_THREADS;
        // (meta is chunk-invariant: hidden from the optimiser, else it hoists every slot's coefficient
        // addresses out of the chunk loop into ~100 registers and halves the occupancy)
        uint32_t m = p.meta[j];
        asm volatile("" : "+v"(m));
        if (e < total) lds[m & 0xffffu] = tensor_transform(t, coef, p.v[j], HASY ? p.y[j] : 0.f, (int)(m >> 16), C);
    }
}

__device__ __forceinline__ void ig_load_coef(const float* g, float* l, int n) {
    if (g != nullptr)
        for (int i = threadIdx.x; i < n; i += IG_THREADS) l[i] = g[i];
}

template <int NT, int TI, bool GRAD, int ND, bool BF = false>
__global__ __launch_bounds__(IG_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void igemm_gather_k(IgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const IgGeom& g = a.g;
    float* lin = smem;
    float* lw = lin + a.lds_in;
    float* lout = lw + a.lds_w;
    int* ktab = reinterpret_cast<int*>(lout + a.lds_out);
    uint32_t* ptab = reinterpret_cast<uint32_t*>(ktab + a.lds_tab);  // per class pixel: in offset | out offset << 16
    float* cin = reinterpret_cast<float*>(ptab + a.lds_ptab);  // [8*Cin] input transform coefficients
    float* cep = cin + 8 * g.Cin;                             // [8*N]   epilogue (producer BN) coefficients
    float* lbias = cep + 8 * g.N;                             // [Np]    bias (zero padded)

    IG_TDECL;
    IG_T();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int HpWp = g.Hp * g.Wp, CinHpWp = g.Cin * HpWp;
    const int OHW = g.OH * g.OW;
    const int n_ntile = g.Np >> 4;

    // the first chunk's loads are issued before anything else: the set-up below hides their latency
    const int CHW = g.Cin * g.H * g.W;
    BnFinRegs finr;
    bnfin_issue<false>(a.fin, finr);  // the producer BatchNorm's accumulators (finalised below, behind the tables)
    PreRegs<IG_PRE, GRAD> pr;
    if (a.pre) {
        pre_decode(pr, (uint32_t)(a.SB * CHW), g.Cin, g.H * g.W, g.W, g.Hp, g.Wp, g.halo_h, g.halo_w, g.d_CHW, g.d_HW,
                   g.d_W);
        if ((int)blockIdx.x < a.nchunks)
            pre_load(a.in, (int64_t)blockIdx.x * a.SB * CHW, (uint32_t)(min(a.SB, a.B - (int)blockIdx.x * a.SB) * CHW),
                     pr);
    }
    // ---- once per workgroup (igemm_tables.inc): clear the input image, load the coefficient blocks and the bias,
    // build the pixel table, the tap-offset tables and the [k][n] weight images of every class
    float* const dummy = lbias + g.Np;  // spare LDS word: masked-off epilogue stores land here
    int* const wbase = reinterpret_cast<int*>(lbias + g.Np + 8);  // [lds_tab] weight base index per reduction row (set-up only)
    // [4*Np] this workgroup's running per-channel sums over its chunks (one slab row per WORKGROUP, written at the end)
    float* const wstat = reinterpret_cast<float*>(wbase + a.lds_tab);
    for (int i = tid; i < 4 * g.Np; i += IG_THREADS) wstat[i] = 0.f;
    // the producer's BatchNorm finalised here (every workgroup merges the producer's rows; lout is free until the first
    // tiles): behind the first chunk's loads, before the tables
    const float* const cin_src = a.fin.C > 0 ? nullptr : a.in.coef;
    if (a.ncls > 1) {
#define MULTI 1
#include "igemm_tables.inc"
#undef MULTI
    } else {
#define MULTI 0
#include "igemm_tables.inc"
#undef MULTI
    }
    if (a.fin.C > 0) bnfin_forward<IG_THREADS>(a.fin, finr, cin, lout);  // (lout is free until the first tiles)

    // Chunk loop, two LDS-only barriers per round:
    //   [barrier] issue the loads of chunk c+1 | MFMA tiles of chunk c (lin -> lout)
    //   [barrier] stage chunk c+1 into lin     | copy chunk c out of lout (+ statistics)
    // The copy-out's global stores are never waited for by a barrier: they drain under the next chunk's MFMAs.
    // The first round has no chunk to compute (chunk < 0) and only stages the workgroup's first one, so that the
    // staging code exists once (a second inlined copy costs ~100 registers).
    for (int chunk = -1, nx = blockIdx.x; chunk < a.nchunks; chunk = nx, nx += gridDim.x) {
        const int s0 = max(chunk, 0) * a.SB;
        const int sb = min(a.SB, a.B - s0);
        lds_barrier();  // this chunk's image is staged (first round: tables ready); the previous copy-out is done with lout
        IG_T();
        if (chunk >= 0) {
        if (a.pre && nx < a.nchunks)  // the next chunk's loads fly while this one is computed
            pre_load(a.in, (int64_t)nx * a.SB * CHW, (uint32_t)(min(a.SB, a.B - nx * a.SB) * CHW), pr);

        // ---- output tiles: 16 pixels of one class x 16*NT channels, round-robin over the waves
        for (int cls = 0; cls < a.ncls; ++cls) {
            const IgClass& c = a.cls[cls];
            const int AHW = c.AH * c.AW;
            const int P = sb * AHW;
            if constexpr (ND > 0) {
                // Thin layers (<= ND output channels, short K): 16x16 MFMA tiles would be mostly padding and the
                // fp32 matrix pipe the bottleneck, so every lane takes one output pixel on the vector ALU; the
                // tap offsets and weights are wave-uniform LDS broadcasts.
                for (int q = tid; q < P; q += IG_THREADS) {
                    const uint32_t pe = ptab[c.poff + q];
                    const float* lp = lin + (pe & 0xffffu);
                    float* op = lout + (pe >> 16);
                    float acc[ND];
#pragma unroll
                    for (int n = 0; n < ND; ++n) acc[n] = lbias[n];
                    const int* kt = ktab + c.toff;
                    const float* wk = lw + c.woff;
                    // four taps per iteration, every LDS read of the group issued before its first FMA; the
                    // rows between Kc and its multiple of 4 hold zero weights and in-range offsets
                    for (int k = 0; k < c.Kc; k += 4) {
                        const int o0 = kt[k], o1 = kt[k + 1], o2 = kt[k + 2], o3 = kt[k + 3];
                        const float a0 = lp[o0], a1 = lp[o1], a2 = lp[o2], a3 = lp[o3];
                        const float* w0 = wk + k * g.Np;
                        if constexpr (ND == 8) {  // lw is 16-byte aligned (host pads lds_in), rows are 16 floats
                            const float4* w4 = reinterpret_cast<const float4*>(w0);
                            const int rs = g.Np >> 2;
                            const float4 p0 = w4[0], p1 = w4[1], q0 = w4[rs], q1 = w4[rs + 1];
                            const float4 r0 = w4[2 * rs], r1 = w4[2 * rs + 1], t0 = w4[3 * rs], t1 = w4[3 * rs + 1];
                            acc[0] = fmaf(a0, p0.x, fmaf(a1, q0.x, fmaf(a2, r0.x, fmaf(a3, t0.x, acc[0]))));
                            acc[1] = fmaf(a0, p0.y, fmaf(a1, q0.y, fmaf(a2, r0.y, fmaf(a3, t0.y, acc[1]))));
                            acc[2] = fmaf(a0, p0.z, fmaf(a1, q0.z, fmaf(a2, r0.z, fmaf(a3, t0.z, acc[2]))));
                            acc[3] = fmaf(a0, p0.w, fmaf(a1, q0.w, fmaf(a2, r0.w, fmaf(a3, t0.w, acc[3]))));
                            acc[4] = fmaf(a0, p1.x, fmaf(a1, q1.x, fmaf(a2, r1.x, fmaf(a3, t1.x, acc[4]))));
                            acc[5] = fmaf(a0, p1.y, fmaf(a1, q1.y, fmaf(a2, r1.y, fmaf(a3, t1.y, acc[5]))));
                            acc[6] = fmaf(a0, p1.z, fmaf(a1, q1.z, fmaf(a2, r1.z, fmaf(a3, t1.z, acc[6]))));
                            acc[7] = fmaf(a0, p1.w, fmaf(a1, q1.w, fmaf(a2, r1.w, fmaf(a3, t1.w, acc[7]))));
                        } else {
#pragma unroll
                            for (int n = 0; n < ND; ++n)
                                acc[n] = fmaf(a0, w0[n], fmaf(a1, w0[g.Np + n], fmaf(a2, w0[2 * g.Np + n], fmaf(a3, w0[3 * g.Np + n], acc[n]))));
                        }
                    }
#pragma unroll
                    for (int n = 0; n < ND; ++n)
                        if (n < g.N) op[n * OHW] = acc[n];
                }
                continue;
            }
            const int ntile = (P + 15) >> 4;
            // TI tiles in flight per wave: their table -> gather -> MFMA -> store chains are independent,
            // which is what hides the LDS and MFMA latencies (the kernels are latency-, not issue-bound)
            // the classes of a transposed gather run side by side: each owns waves [w0, w0 + nw) (host: in proportion
            // to its tiles x K), instead of all eight waves walking the classes one after the other
            if (wid < c.w0 || wid >= c.w0 + c.nw) continue;
            // one pass = TT tiles of this wave in flight; the last pass of a wave may have only one valid tile, and
            // then runs the single-tile instance instead of spending matrix-pipe time on a masked one
            auto pass = [&](auto tt_c, const int t0) {
                constexpr int TT = decltype(tt_c)::value;
                const float* lp[TT];
                int outoff[TT];
#pragma unroll
                for (int i = 0; i < TT; ++i) {
                    const int q = (t0 + i * c.nw) * 16 + li;
                    const bool valid = q < P;
                    const uint32_t pe = ptab[c.poff + (valid ? q : 0)];
                    lp[i] = lin + (pe & 0xffffu);
                    outoff[i] = valid ? (int)(pe >> 16) : -1;
                }
                const int* kt = ktab + c.toff + lk;
                for (int nc = 0; nc < n_ntile; nc += NT) {
                    f32x4 acc[TT][NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < TT; ++i) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (BF) {  // one bf16 MFMA per 16 k-values: this lane's are k4 + lk*4 + 0..3
                        const int* ktb = ktab + c.toff + lk * 4;
                        const float* lwb = lw + c.woff + lk * 4 * g.Np + nc * 16 + li;
                        for (int k4 = 0; k4 < c.KcPad; k4 += 16) {
                            const int o0 = ktb[k4], o1 = ktb[k4 + 1], o2 = ktb[k4 + 2], o3 = ktb[k4 + 3];
                            s16x4 av[TT], bw[NT];
#pragma unroll
                            for (int i = 0; i < TT; ++i) av[i] = bf16_pack4(lp[i][o0], lp[i][o1], lp[i][o2], lp[i][o3]);
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                const float* wj = lwb + k4 * g.Np + (nc + j < n_ntile ? j * 16 : 0);
                                bw[j] = bf16_pack4(wj[0], wj[g.Np], wj[2 * g.Np], wj[3 * g.Np]);
                            }
#pragma unroll
                            for (int i = 0; i < TT; ++i)
#pragma unroll
                                for (int j = 0; j < NT; ++j) acc[i][j] = MFMA_BF16(bw[j], av[i], acc[i][j]);
                        }
                    } else {
                    const float* lwc = lw + c.woff + lk * g.Np + nc * 16 + li;
                    // KcPad is a multiple of 16: four independent k-steps per iteration
                    for (int k4 = 0; k4 < c.KcPad; k4 += 16) {
                        const int o0 = kt[k4], o1 = kt[k4 + 4], o2 = kt[k4 + 8], o3 = kt[k4 + 12];
                        float av[TT][4], bw[NT][4];
#pragma unroll
                        for (int i = 0; i < TT; ++i) {
                            av[i][0] = lp[i][o0]; av[i][1] = lp[i][o1]; av[i][2] = lp[i][o2]; av[i][3] = lp[i][o3];
                        }
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            // (a padding channel tile reads finite weights of the next class / table: harmless)
                            const float* wj = lwc + k4 * g.Np + (nc + j < n_ntile ? j * 16 : 0);
                            bw[j][0] = wj[0]; bw[j][1] = wj[4 * g.Np]; bw[j][2] = wj[8 * g.Np]; bw[j][3] = wj[12 * g.Np];
                        }
#pragma unroll
                        for (int h = 0; h < 4; ++h)
#pragma unroll
                            for (int i = 0; i < TT; ++i)
#pragma unroll
                                for (int j = 0; j < NT; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[j][h], av[i][h], acc[i][j], 0, 0, 0);
                    }
                    }  // fp32 operands
                    // A = weights, B = pixels: D row (channel) = lk*4 + r, col (pixel) = li, so every lane
                    // stores the four channels of its own pixel (no cross-lane traffic in the epilogue)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int n0 = (nc + j) * 16 + lk * 4;
                        if (n0 < g.N) {  // (false for a whole padding tile: n0 >= Np)
                            const float b0 = lbias[n0], b1 = lbias[n0 + 1], b2 = lbias[n0 + 2], b3 = lbias[n0 + 3];
                            const bool full4 = n0 + 3 < g.N;
#pragma unroll
                            for (int i = 0; i < TT; ++i) {
                                // branch-free: a masked pixel writes its four values to the spare word (stride 0)
                                const bool ok = outoff[i] >= 0;
                                float* base = ok ? lout + outoff[i] + n0 * OHW : dummy;
                                const int st = ok ? OHW : 0;
                                if (full4) {
                                    base[0] = acc[i][j][0] + b0;
                                    base[st] = acc[i][j][1] + b1;
                                    base[2 * st] = acc[i][j][2] + b2;
                                    base[3 * st] = acc[i][j][3] + b3;
                                } else {  // channel count not a multiple of 4: per-channel mask
                                    base[0] = acc[i][j][0] + b0;
                                    *(n0 + 1 < g.N ? base + st : dummy) = acc[i][j][1] + b1;
                                    *(n0 + 2 < g.N ? base + 2 * st : dummy) = acc[i][j][2] + b2;
                                    *(n0 + 3 < g.N ? base + 3 * st : dummy) = acc[i][j][3] + b3;
                                }
                            }
                        }
                    }
                }
            };
            for (int t0 = wid - c.w0; t0 < ntile; t0 += c.nw * TI) {
                if (TI > 1 && t0 + c.nw >= ntile) pass(std::integral_constant<int, 1>{}, t0);
                else pass(std::integral_constant<int, TI>{}, t0);
            }
        }
        }  // chunk >= 0
        IG_T();
        lds_barrier();  // lout is complete, lin is free
        IG_T();
        if (nx < a.nchunks) {
            const int cs0 = nx * a.SB, csb = min(a.SB, a.B - cs0);
            if (a.pre)
                pre_store(a.in, cin, lin, (uint32_t)(csb * CHW), g.Cin, pr);
            else
                ig_stage(a.in, cin, lin, (int64_t)cs0 * CHW, csb, g.Cin, g.H * g.W, g.W, g.Hp, g.Wp, g.halo_h, g.halo_w,
                         g.d_CHW, g.d_HW, g.d_W);
        }
        IG_T();
        if (chunk < 0) continue;

        // ---- write the chunk out (contiguous in global memory) + per-channel sums
        const int64_t obase = (int64_t)s0 * g.N * OHW;
        const int total = sb * g.N * OHW;
        if (a.ep_mode == EP_PLAIN) {
            // (the activation runs here, once per real output element with every lane busy, not in the
            // MFMA epilogue where 15 of 16 lanes may hold padding channels)
            if (((g.N * OHW) & 3) == 0) {  // chunk start and length are multiples of 4 floats: 16-byte stores
                const float4* l4 = reinterpret_cast<const float4*>(lout);
                float4* o4 = reinterpret_cast<float4*>(a.out + obase);
                for (int e = tid; e < (total >> 2); e += IG_THREADS) {
                    float4 v = l4[e];
                    v.x = act_apply(v.x, a.act, a.slope); v.y = act_apply(v.y, a.act, a.slope);
                    v.z = act_apply(v.z, a.act, a.slope); v.w = act_apply(v.w, a.act, a.slope);
                    o4[e] = v;
                }
            } else {
                for (int e = tid; e < total; e += IG_THREADS) a.out[obase + e] = act_apply(lout[e], a.act, a.slope);
            }
            if (a.slab == nullptr) continue;
            // BatchNorm statistics from the LDS image, one wave per channel (no global traffic in this loop):
            // sums stay in registers and are reduced once; no atomics, fixed order => deterministic
            for (int n = wid; n < g.N; n += IG_WAVES) {
                const float pv = act_apply(lout[n * OHW], a.act, a.slope);  // chunk-local pivot: the channel's first value
                float s1 = 0.f, s2 = 0.f;
                for (int s = 0; s < sb; ++s) {
                    const int rbase = (s * g.N + n) * OHW;
                    for (int i = lane; i < OHW; i += 64) {
                        const float d = act_apply(lout[rbase + i], a.act, a.slope) - pv;
                        s1 += d;
                        s2 = fmaf(d, d, s2);
                    }
                }
                s1 = wave_sum(s1);
                s2 = wave_sum(s2);
                if (lane == 0) {  // Chan-merge this chunk into the workgroup's running (n, mean, M2) of the channel
                    const float cn = (float)(sb * OHW), dd = s1 / cn;
                    float rn = wstat[4 * n], rmean = wstat[4 * n + 1], rm2 = wstat[4 * n + 2];
                    chan_merge(rn, rmean, rm2, cn, pv + dd, fmaxf(s2 - s1 * dd, 0.f));
                    wstat[4 * n] = rn; wstat[4 * n + 1] = rmean; wstat[4 * n + 2] = rm2;
                }
            }
            continue;
        }
        // EP_BNGRAD (backward through the producer's virtual BatchNorm): one wave per channel walks that channel's
        // row in every sample of the chunk, so the two sums stay in registers and are reduced once; no atomics,
        // fixed order => deterministic.  The producer outputs y come from global memory: the loads of eight row
        // segments are issued together (one at a time, each exposed its full latency: 3-5 us per chunk).
        const int nI = (OHW + 63) >> 6, nIt = sb * nI;
        for (int n = wid; n < g.N; n += IG_WAVES) {
            const float mean = cep[n], rstd = cep[g.N + n], sc = cep[2 * g.N + n], sh = cep[3 * g.N + n];
            float s1 = 0.f, s2 = 0.f;
            for (int it0 = 0; it0 < nIt; it0 += 8) {
                float yv[8];
                int off[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int it = it0 + u, s = it / nI, i = (it - s * nI) * 64 + lane;
                    off[u] = (it < nIt && i < OHW) ? (s * g.N + n) * OHW + i : -1;
                    yv[u] = off[u] >= 0 ? a.ep_y[obase + off[u]] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (off[u] >= 0) {
                        float v = act_apply(lout[off[u]], a.act, a.slope);
                        if (a.ep_relu && fmaf(yv[u] - mean, sc, sh) <= 0.f) v = 0.f;
                        s1 += v;
                        s2 = fmaf(v, (yv[u] - mean) * rstd, s2);
                        a.out[obase + off[u]] = v;
                    }
            }
            if (a.slab != nullptr) {
                s1 = wave_sum(s1);
                s2 = wave_sum(s2);
                if (lane == 0) {
                    wstat[4 * n] += s1;
                    wstat[4 * n + 1] += s2;
                }
            }
        }
    }
    // one slab row per workgroup (channel n belongs to wave n % IG_WAVES all along: no barrier needed)
    if (a.slab != nullptr && lane == 0)
        for (int n = wid; n < g.N; n += IG_WAVES) {
            float4* sl = reinterpret_cast<float4*>(a.slab + ((int64_t)blockIdx.x * g.N + n) * 4);
            if (a.ep_mode == EP_PLAIN) {
                *sl = make_float4(wstat[4 * n + 1], 0.f, wstat[4 * n + 2], wstat[4 * n]);  // {pivot = mean, 0, M2, n}
                if (a.acc_save != nullptr) {
                    const float pv = a.acc_pivot[n];
                    bnacc_add_fwd(a.acc_save, g.N, n, wstat[4 * n], wstat[4 * n + 1], wstat[4 * n + 2], pv);
                    if (blockIdx.x == 0) a.acc_save[7 * g.N + n] = pv;
                }
            } else {
                *sl = make_float4(wstat[4 * n], wstat[4 * n + 1], 0.f, 0.f);
                if (a.acc_save != nullptr) bnacc_add_bwd(a.acc_save, g.N, n, wstat[4 * n], wstat[4 * n + 1]);
            }
        }
    IG_T();
#ifdef IG_TIMERS
    if (threadIdx.x == 0 && blockIdx.x == 1)
        printf("gather<%d,%d,%d,%d> Cin %d N %d H %d OH %d SB %d chunks %d grid %d: setup | per round: [top barrier] tiles, barrier, stage(next), copy-out+stats ...", NT, TI, (int)GRAD, ND,
               g.Cin, g.N, g.H, g.OH, a.SB, a.nchunks, (int)gridDim.x);
#endif
    IG_TPRINT("");
}

// ---- weight gradient --------------------------------------------------------------------------
// dw[m][c][kh][kw] += sum_{b,oh,ow} S[b,m,oh,ow] * G[b,c,oh*sh-ph+kh,ow*sw-pw+kw]
//   Conv : S = d(pre-activation) [Cout], G = layer input [Cin];  ConvT: S = layer input, G = d(pre-act).
// MFMA: A[m][k=pixel] from the S tile, B[k=pixel][n=(c,tap)] gathered from the padded G tile.
// Each workgroup loops over its chunks keeping ALL (m,n) accumulator tiles (<= WG_MAXT) in
// registers; its waves split the pixels.  One cross-wave LDS reduction and one atomicAdd per
// element happen per WORKGROUP, not per chunk.  The bias gradient is the per-channel sum of the
// d(pre-activation) tensor, taken from its LDS image (a zero halo adds nothing).
#define WG_MAXT 20
#define WG_SMALL_T 4  // kernels with at most this many accumulator tiles run at two workgroups per CU (<= 128 VGPRs; the 2x2 instance spills 31 registers there and is still faster than one workgroup per CU)

struct WgArgs {
    TensorIn S, G;
    float* part;      // [gridDim.x][MT*NT*256 + 64] workgroup partials (reduced by wgrad_finalize_k)
    int want_bias;
    int pre;  // 1: chunks fit the register prefetch (<= WG_PRE*IG_THREADS elements of S and of G)
    int bias_from_S;  // 1: dbias from S (Conv), 0: from G (ConvT)
    int B, SB, nchunks;
    int Sc, SH, SW;   // S: channels, small grid
    int Gc, GH, GW;   // G: channels, big grid
    int KH, KW, sh, sw, ph, pw;
    int Hp, Wp;       // padded G image (halo = ph, pw low side)
    FastDiv dS_HW, dS_W, dS_CHW, dG_HW, dG_W, dG_CHW, d_KK, d_KW, d_bC;
    int lds_S, lds_G, lds_tab;
};

// MT x NT accumulator tiles (16x16 each) per wave: compile-time so that every operand read of a step
// is issued before its first MFMA and the loop has no branches.  Rows m >= Sc and columns
// n >= Gc*KH*KW read a clamped (valid) row/column; their results are simply never written.
template <int MT, int NT, bool BF = false>
__global__ __launch_bounds__(IG_THREADS) __attribute__((amdgpu_waves_per_eu(MT * NT <= WG_SMALL_T ? 4 : 2, MT * NT <= WG_SMALL_T ? 4 : 2))) void igemm_wgrad_k(WgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lS = smem;                 // [sb][Sc][SHW]
    float* lG = lS + a.lds_S;         // [sb][Gc][Hp][Wp]
    int2* ptab = reinterpret_cast<int2*>(lG + a.lds_G);  // per pixel: (S offset, G base)
    float* cS = reinterpret_cast<float*>(ptab + a.lds_tab);  // [8*Sc] transform coefficients of S
    float* cG = cS + 8 * a.Sc;                                // [8*Gc] ... of G
    float* bred = cG + 8 * a.Gc;                              // [IG_WAVES][64] bias partials

    IG_TDECL;
    IG_T();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int SHW = a.SH * a.SW, HpWp = a.Hp * a.Wp, KK = a.KH * a.KW;
    const int NK = a.Gc * KK;
    const int Pfull = a.SB * SHW;

    // the first chunk's loads are issued before anything else: the set-up below hides their latency
    const int SCHW = a.Sc * SHW, GCHW = a.Gc * a.GH * a.GW;
    PreRegs<WG_PRE_OF(MT * NT), true> pS, pG;
    if (a.pre) {
        pre_decode(pS, (uint32_t)(a.SB * SCHW), a.Sc, SHW, a.SW, a.SH, a.SW, 0, 0, a.dS_CHW, a.dS_HW, a.dS_W);
        pre_decode(pG, (uint32_t)(a.SB * GCHW), a.Gc, a.GH * a.GW, a.GW, a.Hp, a.Wp, a.ph, a.pw, a.dG_CHW, a.dG_HW,
                   a.dG_W);
    }
    if (a.pre && (int)blockIdx.x < a.nchunks) {
        const int nsb = min(a.SB, a.B - (int)blockIdx.x * a.SB);
        pre_load(a.S, (int64_t)blockIdx.x * a.SB * SCHW, (uint32_t)(nsb * SCHW), pS);
        pre_load(a.G, (int64_t)blockIdx.x * a.SB * GCHW, (uint32_t)(nsb * GCHW), pG);
    }
    for (int i = tid; i < a.lds_G; i += IG_THREADS) lG[i] = 0.f;  // halo stays zero for every chunk
    ig_load_coef(a.S.coef, cS, 8 * a.Sc);
    ig_load_coef(a.G.coef, cG, 8 * a.Gc);
    // pixel table (same for every chunk): q -> S offset (s*Sc*SHW + pix), G base (s*Gc*HpWp + oh*sh*Wp + ow*sw)
    for (int q = tid; q < Pfull; q += IG_THREADS) {
        const int s = fdiv((uint32_t)q, a.dS_HW), pix = q - s * SHW;
        const int oh = fdiv((uint32_t)pix, a.dS_W), ow = pix - oh * a.SW;
        ptab[q] = make_int2(s * a.Sc * SHW + pix, s * a.Gc * HpWp + oh * a.sh * a.Wp + ow * a.sw);
    }

    // per-lane constants: A row m = mt*16 + li -> channel offset in the S image;
    // B column n = nt*16 + li -> (channel, tap) offset in the G image
    int moff[MT], koffN[NT];
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) moff[mt] = min(mt * 16 + li, a.Sc - 1) * SHW;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = min(nt * 16 + li, NK - 1);
        const int c = fdiv((uint32_t)n, a.d_KK), tap = n - c * KK;
        const int kh = fdiv((uint32_t)tap, a.d_KW), kw = tap - kh * a.KW;
        koffN[nt] = c * HpWp + kh * a.Wp + kw;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float bsum = 0.f;  // lane c of every wave: this wave's share of the bias gradient of channel c (< 64)

    for (int chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x) {
        const int s0 = chunk * a.SB;
        const int sb = min(a.SB, a.B - s0);
        const int P = sb * SHW;
        __syncthreads();  // tables ready / previous chunk's MFMAs are done with the tiles
        IG_T();
        if (a.pre) {
            pre_store(a.S, cS, lS, (uint32_t)(sb * SCHW), a.Sc, pS);
            pre_store(a.G, cG, lG, (uint32_t)(sb * GCHW), a.Gc, pG);
        } else {
            ig_stage(a.S, cS, lS, (int64_t)s0 * SCHW, sb, a.Sc, SHW, a.SW, a.SH, a.SW, 0, 0, a.dS_CHW, a.dS_HW,
                     a.dS_W);
            ig_stage(a.G, cG, lG, (int64_t)s0 * GCHW, sb, a.Gc, a.GH * a.GW, a.GW, a.Hp, a.Wp, a.ph, a.pw, a.dG_CHW,
                     a.dG_HW, a.dG_W);
        }
        __syncthreads();
        IG_T();
        {  // the next chunk's loads fly while this one is computed
            const int nx = chunk + gridDim.x;
            if (a.pre && nx < a.nchunks) {
                const int nsb = min(a.SB, a.B - nx * a.SB);
                pre_load(a.S, (int64_t)nx * a.SB * SCHW, (uint32_t)(nsb * SCHW), pS);
                pre_load(a.G, (int64_t)nx * a.SB * GCHW, (uint32_t)(nsb * GCHW), pG);
            }
        }
        // bias gradient = per-channel sum of the d(pre-activation) image (its zero halo adds nothing):
        // the (sample, channel) rows go round-robin over the waves
        if (a.want_bias) {
            const float* img = a.bias_from_S ? lS : lG;
            const int C = a.bias_from_S ? a.Sc : a.Gc;
            const int rowlen = a.bias_from_S ? SHW : HpWp;
            for (int u = wid; u < sb * C; u += IG_WAVES) {
                const float* row = img + u * rowlen;
                float t = 0.f;
                for (int i = lane; i < rowlen; i += 64) t += row[i];
                t = wave_sum(t);
                const int c = u - (int)fdiv((uint32_t)u, a.d_bC) * C;
                if (lane == c) bsum += t;
            }
        }
        if constexpr (BF) {  // sixteen pixels per bf16 MFMA: this lane's are q16 + lk*4 + 0..3
            for (int q16 = wid * 16; q16 < P; q16 += 16 * IG_WAVES) {
                const int q0 = q16 + lk * 4;
                int2 pp[4];
                bool pv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pv[j] = q0 + j < P;
                    pp[j] = ptab[pv[j] ? q0 + j : 0];
                }
                s16x4 av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {  // pixels past the chunk: zero A column, finite B
                    const float s0 = lS[pp[0].x + moff[mt]], s1 = lS[pp[1].x + moff[mt]];
                    const float s2 = lS[pp[2].x + moff[mt]], s3 = lS[pp[3].x + moff[mt]];
                    av[mt] = bf16_pack4(pv[0] ? s0 : 0.f, pv[1] ? s1 : 0.f, pv[2] ? s2 : 0.f, pv[3] ? s3 : 0.f);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bv[nt] = bf16_pack4(lG[pp[0].y + koffN[nt]], lG[pp[1].y + koffN[nt]], lG[pp[2].y + koffN[nt]],
                                        lG[pp[3].y + koffN[nt]]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = MFMA_BF16(av[mt], bv[nt], acc[mt][nt]);
            }
        } else
        // pixels: four per MFMA step, two independent steps per iteration, steps round-robin over the waves
        for (int q4 = wid * 4; q4 < P; q4 += 8 * IG_WAVES) {
            const int qa = q4 + lk, qb = qa + 4 * IG_WAVES;
            const bool va = qa < P, vb = qb < P;
            const int2 pa = ptab[va ? qa : 0], pb = ptab[vb ? qb : 0];
            float av[2][MT], bv[2][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                av[0][mt] = lS[pa.x + moff[mt]];
                av[1][mt] = lS[pb.x + moff[mt]];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bv[0][nt] = lG[pa.y + koffN[nt]];
                bv[1][nt] = lG[pb.y + koffN[nt]];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {  // pixels past the chunk: zero A column, finite B
                av[0][mt] = va ? av[0][mt] : 0.f;
                av[1][mt] = vb ? av[1][mt] : 0.f;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][mt], bv[h][nt], acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();
    IG_T();
    // cross-wave reduction in LDS (rbuf overlays the dead S/G tiles; the host sizes them >= MT*NT*256);
    // the workgroup's partial goes to its own row of `part` in accumulator-tile order, bias sums behind it
    // (global atomics on the few cache lines of dw serialise in L2 and cost more than the whole MFMA loop)
    constexpr int TE = MT * NT * 256;
    constexpr int TPT = (TE + IG_THREADS - 1) / IG_THREADS;
    const int nw = min(IG_WAVES, (a.lds_S + a.lds_G) / TE);  // waves whose tiles fit in LDS at once (>= 1)
    float tot[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) tot[k] = 0.f;
    bred[wid * 64 + lane] = bsum;
    for (int w0 = 0; w0 < IG_WAVES; w0 += nw) {
        if (wid >= w0 && wid < w0 + nw) {
            float* slot = smem + (wid - w0) * TE;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slot[((mt * NT + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
        }
        __syncthreads();
        const int cnt = min(nw, IG_WAVES - w0);
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int i = tid + k * IG_THREADS;
            if (i < TE)
                for (int w = 0; w < cnt; ++w) tot[k] += smem[w * TE + i];
        }
        __syncthreads();
    }
    float* row = a.part + (int64_t)blockIdx.x * (TE + 64);
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = tid + k * IG_THREADS;
        if (i < TE) row[i] = tot[k];
    }
    if (tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < IG_WAVES; ++w) t += bred[w * 64 + tid];
        row[MT * NT * 256 + tid] = t;
    }
    IG_T();
#ifdef IG_TIMERS
    if (threadIdx.x == 0 && blockIdx.x == 1)
        printf("wgrad<%d,%d> Sc %d Gc %d SH %d GH %d SB %d chunks %d grid %d: setup | per chunk: stage mfma(+bias) | ... reduce:", MT, NT, a.Sc, a.Gc,
               a.SH, a.GH, a.SB, a.nchunks, (int)gridDim.x);
#endif
    IG_TPRINT("");
}

// Sum the workgroup partials of igemm_wgrad_k and add them into dw / dbias, for EVERY conv layer of a backward
// pass in one launch (each layer keeps its own partial region; the sums only have to exist before the
// optimiser).  One workgroup per 16 partial columns of a layer; its 16 slices stride over the partial rows,
// eight independent loads in flight each.  Flat grid: a workgroup finds its layer from first_block.
#define WG_FIN_MAX_LAYERS 12
struct WgFinLayer {
    const float* part;
    float* dw;
    float* dbias;
    int nrows, NT, T, Sc, NK, C;
    int plain;        // 1: rows are [dw (Sc*NK floats, in dw's own order)][dbias (C floats)] (Linear layers); 0: MFMA tile order
    int stride;       // floats per partial row
    int ncols;        // columns to reduce (dw + bias sums)
    int first_block;  // the layer owns ceil(ncols / 16) workgroups from here
};
struct WgFinMulti {
    WgFinLayer l[WG_FIN_MAX_LAYERS];
    int n;
    // elbo_fused_k's per-workgroup {recon, kld} pairs, summed by ONE EXTRA workgroup (the grid's last) in
    // elbo_finalize_k's order into out[0..3]; nullptr: none pending
    const float* loss_part;
    float* loss_out;
    int loss_blocks, loss_accumulate;
};
__global__ __launch_bounds__(256) void wgrad_finalize_k(WgFinMulti m) {
    __shared__ float red[16][17];
    if (m.loss_part != nullptr && blockIdx.x == gridDim.x - 1) {
        float r = 0.f, k = 0.f;
        for (int i = threadIdx.x; i < m.loss_blocks; i += blockDim.x) {
            r += m.loss_part[2 * i];
            k += m.loss_part[2 * i + 1];
        }
        r = block_sum_256(r, &red[0][0]);
        k = block_sum_256(k, &red[0][0]);
        if (threadIdx.x == 0) {
            m.loss_out[0] = r + k;
            m.loss_out[1] = r;
            m.loss_out[2] = k;
            if (m.loss_accumulate) m.loss_out[3] += r + k;
        }
        return;
    }
    int layer = 0;
#pragma unroll
    for (int i = 1; i < WG_FIN_MAX_LAYERS; ++i)
        if (i < m.n && (int)blockIdx.x >= m.l[i].first_block) layer = i;
    const WgFinLayer& L = m.l[layer];
    const float* __restrict__ part = L.part;
    const int T = L.T, NT = L.NT, nrows = L.nrows;
    const int stride = L.stride;
    const int jj = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int j = ((int)blockIdx.x - L.first_block) * 16 + jj;
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = 0.f;
    int b = j < L.ncols ? slice : nrows;  // (the last workgroup of a layer may reach past its columns)
    for (; b + 7 * 16 < nrows; b += 8 * 16)
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] += part[(int64_t)(b + u * 16) * stride + j];
    for (; b < nrows; b += 16) t[0] += part[(int64_t)b * stride + j];
    red[slice][jj] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    __syncthreads();
    if (slice != 0) return;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += red[k][jj];
    if (L.plain) {
        if (j < L.Sc * L.NK) L.dw[j] += sum;
        else if (L.dbias != nullptr && j - L.Sc * L.NK < L.C) L.dbias[j - L.Sc * L.NK] += sum;
    } else if (j < T * 256) {
        const int tile = j >> 8, r = (j >> 6) & 3, l = j & 63;
        const int mt = tile / NT, nt = tile - mt * NT;
        const int mrow = mt * 16 + (l >> 4) * 4 + r, n = nt * 16 + (l & 15);
        if (mrow < L.Sc && n < L.NK) L.dw[(int64_t)mrow * L.NK + n] += sum;
    } else if (L.dbias != nullptr && j - T * 256 < L.C) {
        L.dbias[j - T * 256] += sum;
    }
}

// ---- virtual BatchNorm finalize ------------------------------------------------------------
// One wave per channel.  forward: every block left {pivot_b, sum(y-p_b), sum((y-p_b)^2), n_b}; each is
// turned into (n, mean, M2) and merged with Chan's parallel-variance formula, lanes first striding
// over the blocks and then in a fixed butterfly, so the result is deterministic and does not lose
// precision when |mean| >> std.  Writes save = {mean, rstd, scale, beta} and the running stats.
#define BNV_THREADS 256
__global__ __launch_bounds__(BNV_THREADS) void bnv_finalize_fwd_k(const float* __restrict__ slab, int nblocks,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float* __restrict__ save, float* __restrict__ rmean,
                                                                  float* __restrict__ rvar, int C, float eps,
                                                                  float momentum, int training,
                                                                  float* __restrict__ row_out) {
    __shared__ float part[BNV_THREADS / 64][3];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float mean, var;
    if (training) {
        float n = 0.f, mu = 0.f, m2 = 0.f;
        for (int k = tid; k < nblocks; k += BNV_THREADS) {
            const float4 sl = *reinterpret_cast<const float4*>(slab + ((int64_t)k * C + c) * 4);
            if (sl.w > 0.f) {
                const float d = sl.y / sl.w;
                chan_merge(n, mu, m2, sl.w, sl.x + d, fmaxf(sl.z - sl.y * d, 0.f));
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float nb = __shfl_xor(n, off, 64), mb = __shfl_xor(mu, off, 64), m2b = __shfl_xor(m2, off, 64);
            // merge in a fixed (lower lane first) order so both partners compute identical results
            if (lane & off) {
                float tn = nb, tm = mb, t2 = m2b;
                chan_merge(tn, tm, t2, n, mu, m2);
                n = tn; mu = tm; m2 = t2;
            } else {
                chan_merge(n, mu, m2, nb, mb, m2b);
            }
        }
        if (lane == 0) { part[wid][0] = n; part[wid][1] = mu; part[wid][2] = m2; }
        __syncthreads();
        if (tid != 0) return;
#pragma unroll
        for (int w = 1; w < BNV_THREADS / 64; ++w) chan_merge(n, mu, m2, part[w][0], part[w][1], part[w][2]);
        if (row_out != nullptr) {  // data-parallel: this rank's (n, mean, M2) as one slab row for the all-gather
            *reinterpret_cast<float4*>(row_out + (int64_t)c * 4) = make_float4(mu, 0.f, m2, n);
            return;
        }
        mean = mu;
        var = n > 0.f ? m2 / n : 0.f;
        const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    } else {
        if (tid != 0) return;
        mean = rmean[c];
        var = rvar[c];
    }
    const float rstd = rsqrtf(var + eps);
    save[c] = mean;
    save[C + c] = rstd;
    save[2 * C + c] = gamma[c] * rstd;
    save[3 * C + c] = beta[c];
}
// backward: slab of {sum g, sum g*xhat} -> dgamma, dbeta and (a, b, c) of dy = a*g + b*(y-mean) + c
__global__ __launch_bounds__(BNV_THREADS) void bnv_finalize_bwd_k(const float* __restrict__ slab, int nblocks,
                                                                  float* __restrict__ save, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int C, int64_t n_local,
                                                                  float* __restrict__ row_out,
                                                                  const float* __restrict__ local_row) {
    // single GPU: slab rows are workgroup partials, n = n_local.  Data-parallel, pass 1 (row_out): this
    // rank's sums + its count as one row; pass 2 (local_row): rows are ranks, n = sum of their counts, and
    // dgamma/dbeta stay this rank's LOCAL sums (the gradient all-reduce adds the ranks up).
    __shared__ float part[BNV_THREADS / 64][3];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float s1 = 0.f, s2 = 0.f, cnt = 0.f;
    for (int k = tid; k < nblocks; k += BNV_THREADS) {
        const float4 sl = *reinterpret_cast<const float4*>(slab + ((int64_t)k * C + c) * 4);
        s1 += sl.x;
        s2 += sl.y;
        cnt += sl.z;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
    }
    if (lane == 0) { part[wid][0] = s1; part[wid][1] = s2; part[wid][2] = cnt; }
    __syncthreads();
    if (tid != 0) return;
#pragma unroll
    for (int w = 1; w < BNV_THREADS / 64; ++w) { s1 += part[w][0]; s2 += part[w][1]; cnt += part[w][2]; }
    if (row_out != nullptr) {
        *reinterpret_cast<float4*>(row_out + (int64_t)c * 4) = make_float4(s1, s2, (float)n_local, 0.f);
        return;
    }
    float n = (float)n_local;
    if (local_row != nullptr) {
        n = cnt;
        dbeta[c] = local_row[c * 4];
        dgamma[c] = local_row[c * 4 + 1];
    } else {
        dbeta[c] = s1;
        dgamma[c] = s2;
    }
    const float m1 = s1 / n, m2 = s2 / n;
    const float rstd = save[C + c], sc = save[2 * C + c];
    save[4 * C + c] = sc;  // dy = sc*(g - m1 - xhat*m2)
    save[5 * C + c] = -sc * m2 * rstd;
    save[6 * C + c] = -sc * m1;
}
