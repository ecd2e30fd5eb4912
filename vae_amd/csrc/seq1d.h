// seq1d.h - the 1-D model families (TCVAE, multiscale_vae) as one launch per BatchNorm-delimited PHASE.
//
// These models are tiny per sample (<= 2,250 activation elements, 28 k MACs) and every layer is followed by a
// train-mode BatchNorm, i.e. a batch-wide barrier.  The per-layer path needs ~80 launches per train step and is
// launch-bound; here one launch covers everything between two barriers, for BOTH encoders at once:
//
//   forward   FE(i)  i = 0..NE-1   conv_i + act of every encoder (input normalised on load by BN_{i-1})
//             FH                   heads (mu|logvar) + reparameterisation + decoder Linear + convT_0 + act
//             FD(j)  j = 1..ND-1   convT_j + act (+ x[:, :, 0] after the last)
//   backward  BD(j)  j = ND-1..1   BN_j backward (coefficients from the sums the previous phase left), act',
//                                  weight/bias gradient, input gradient + its BatchNorm sums
//             BH                   BD(0) + decoder Linear + reparameterisation + heads, all backward
//             BE(i)  i = NE-1..0   the encoders' conv layers, backward
//
// A workgroup owns chunks of SB samples; everything a phase needs of a chunk is staged in LDS once.  Every
// contraction (conv gather, transposed conv, Linear, all weight and input gradients) is the same device routine:
// a GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32) whose operand and result addresses are SEPARABLE,
//     A(m, k) = a_m(m) + a_k(k),   B(k, n) = b_k(k) + b_n(n),   D(m, n) = d_m(m) + d_n(n),
// each index being a two-level (outer, inner) pair with its own strides (SqIx).  BatchNorm statistics are exact
// two-pass sums per chunk, Chan-merged per workgroup, written as one row per workgroup and merged in a fixed order by
// every consumer workgroup in its prologue (no atomics, no inter-workgroup hand-off inside a launch: the kernel
// boundary is the barrier, which on this chip is cheaper than a grid barrier).  Weight gradients accumulate in LDS
// over a workgroup's chunks and leave as one partial row per workgroup; sq_reduce_k sums the rows in a fixed order.
#pragma once
#include "common.h"
#include "igemm.h"

#define SQ_MAXN 5        // conv layers per encoder / decoder
#define SQ_MAXENC 2
#define SQ_THREADS 512   // 8 waves: one workgroup per CU (the big phases use > 80 KiB of LDS), two waves per SIMD
#define SQ_WAVES (SQ_THREADS / 64)
#define SQ_MAX_GRID 256  // = rows of the per-workgroup statistics; sq_bn_prepare keeps 16 rows per lane in registers
#define SQ_MAXC 64       // channels of any layer
#define SQ_MAXBLK 24     // parameter blocks a phase accumulates gradients for
#define SQ_KT(K) (2 * (((K) + 15) & ~15))  // floats of a K table (int2 entries, padded to 16 rows)
#ifdef SQ_TIMERS  // diagnostic build only: phase stamps of workgroup 1 (10 ns ticks) + shader-clock cycles of the whole span
#define SQ_T(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 1) { tm_[i] = wall_clock64(); ck_[i] = clock64(); } } while (0)
#define SQ_TDECL long long tm_[16] = {0}, ck_[16] = {0}
#define SQ_TPRINT(name, n) do { if (threadIdx.x == 0 && blockIdx.x == 1) { printf("%s", name); for (int q_ = 1; q_ < n; ++q_) printf(" %lld", (tm_[q_] - tm_[q_-1]) * 10); printf(" ns; clock %.0f MHz\n", (double)(ck_[n-1] - ck_[0]) / ((tm_[n-1] - tm_[0]) * 10e-3)); } } while (0)
#else
#define SQ_T(i)
#define SQ_TDECL
#define SQ_TPRINT(name, n)
#endif

struct SqLayer {
    int cin, cout, lin, lout, k, has_bn;
    float bn_eps, bn_mom;
    const float* w;      // conv [cout][cin][k], convT [cin][cout][2]
    const float* b;      // [cout]
    const float* gamma;  // BatchNorm weight / bias (has_bn)
    const float* beta;
    float* rmean;        // running statistics (STATE)
    float* rvar;
    float* save;         // [8*cout]: mean, rstd, gamma*rstd, beta (written by the consumer phase's workgroup 0)
    float* y;            // [B][cout][lout] activated output, before BatchNorm
    float* g;            // [B][cout][lout] gradient w.r.t. the BatchNorm output (w.r.t. y when !has_bn)
    float* fstat;        // [3][cout][grid] per-workgroup {n, mean, M2}
    float* bstat;        // [2][cout][grid] per-workgroup {sum g, sum g*xhat}
    // data parallel (SqDesc.world > 1): one row per rank of this layer's statistics inside SqDesc.gath, gathered
    // between the phases: forward {n, mean, M2}[cout] at gath[fg_off + r*fg_stride], backward {sum g, sum g*xhat}[cout]
    // + n at gath[bg_off + r*bg_stride]
    int fg_off, fg_stride, bg_off, bg_stride;
    int64_t w_off, b_off, ga_off, be_off;  // float offsets of the gradients in the flat GRAD buffer / a partial row
};
struct SqHead {
    const float *wmu, *bmu, *wlv, *blv;  // [lat][K], [lat]
    int64_t wmu_off, bmu_off, wlv_off, blv_off;
    int K;                               // = cout*lout of the encoder's last layer
};
struct SqDesc {
    int B, SB, grid, nchunks, training;
    int prune;       // eval forward without a criterion: the decoder is evaluated at time index 0 only (sq_fh_k)
    int world, rank; // data parallel: BatchNorm statistics over the batch of every rank (rows gathered between phases)
    float* gath;     // the ranks' rows (SqLayer.fg_off / bg_off)
    int NE, ND, nenc, lat, lat_total;
    int C, L;        // input window (C, L)
    float slope;     // LeakyReLU slope of every conv / convT (0 = ReLU)
    SqLayer enc[SQ_MAXENC][SQ_MAXN];
    SqLayer dec[SQ_MAXN];
    SqHead head[SQ_MAXENC];
    const float *wl, *bl;  // decoder Linear [c0][lat_total]
    int64_t wl_off, bl_off;
    const float* x;        // [B][C][L]
    const float* eps;      // [B][lat_total]
    float *mu, *lv, *z, *h0;  // [B][lat_total] x3, [B][c0]
    float* xrec;           // [B][C]
    const float* dxrec;    // [B][C] incoming gradient
    const float *gmu, *glv;  // incoming gradients of mu / logvar (autograd path) or nullptr
    const float* gscale;   // scale of the analytic KL gradient (device float) or nullptr (= 1)
    int kl_grad;
    float* part;           // [grid][nflat] weight-gradient partial rows
    int64_t nflat;
    // criterion fused into the last forward phase (train.py:39-48 / notebook :141-151) when the program ends with ELBO
    int has_elbo, bce, t_sb, t_se;
    const float* target;   // element (b, j) at target[b*t_sb + j*t_se]
    const float* egscale;  // scale of d(loss)/d(x_rec) (device float) or nullptr (= 1)
    float* dxrec_w;        // [B][C] gradient of the loss w.r.t. x_rec, written by the forward
    float* epart;          // [2][grid]... per-workgroup {recon, kld} partial sums: epart[2*wg], epart[2*wg+1]
    // VAE_F_LOSS_IN_BWD: the first launch of the backward pass sums the partials into eout[0..3] (no launch of its own)
    float* eout;
    int efin, eacc;
};

// what elbo_finalize_k does, by workgroup 0 of the first backward launch (fixed order: thread t takes the partials
// t, t + 512, ...; then the block sum)
// sm_fin: the head of the kernel's dynamic LDS (the phase kernels have the whole 160 KiB granted as dynamic LDS, so
// a static array of its own would not fit), free again behind the closing barrier.
__device__ __forceinline__ void sq_elbo_fin(const SqDesc& d, float* sm_fin) {
    if (!d.efin || blockIdx.x != 0) return;
    float r = 0.f, k = 0.f;
    for (int i = threadIdx.x; i < d.grid; i += SQ_THREADS) {
        r += d.epart[2 * i];
        k += d.epart[2 * i + 1];
    }
    r = wave_sum(r);
    k = wave_sum(k);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm_fin[2 * wid] = r; sm_fin[2 * wid + 1] = k; }
    __syncthreads();
    if (threadIdx.x == 0) {
        r = k = 0.f;
        for (int w = 0; w < SQ_WAVES; ++w) { r += sm_fin[2 * w]; k += sm_fin[2 * w + 1]; }
        d.eout[0] = r + k;
        d.eout[1] = r;
        d.eout[2] = k;
        if (d.eacc) d.eout[3] += r + k;
    }
    __syncthreads();
}

// Helpers below are NOT inlined (one copy in the code object, shared by all six phase kernels): a phase executes
// nearly all of its code once per launch, so its time is bounded by instruction fetch, and inlining every GEMM /
// reduction at every call site made kernels of 7-17 k instructions.  Non-inlined functions cannot see that a generic
// pointer is LDS, so they take address-space-3 pointers (SQL casts at the call sites).
typedef __attribute__((address_space(3))) float lds_f;
typedef int __attribute__((ext_vector_type(2))) i32x2;
typedef __attribute__((address_space(3))) i32x2 lds_i2;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
typedef __attribute__((address_space(1))) const float gl_cf;   // global-memory views for the non-inlined helpers
typedef __attribute__((address_space(1))) const f32x4 gl_cf4;
#define SQL(p) ((lds_f*)(p))
#define SQL2(p) ((lds_i2*)(p))
#define SQ_FN __device__ __noinline__

// ---- two-level index -> two offsets -------------------------------------------------------------------------
// idx / d for 0 <= idx < 2^21 by a float reciprocal: (idx + 0.5) / d is at least 0.5 / d away from an integer and the
// computed product is within 2^-22 relative, so the truncation is exact.  One v_rcp_f32 to set up (an integer
// magic-number divisor costs an emulated 32-bit division, and a phase sets up dozens of divisors once per launch).
struct SqDiv {
    float rinv;
    int d;
};
__device__ __forceinline__ SqDiv sq_fdiv(int dd) {
    SqDiv f;
    f.rinv = __builtin_amdgcn_rcpf((float)dd);
    f.d = dd;
    return f;
}
__device__ __forceinline__ uint32_t sq_divi(uint32_t idx, SqDiv f) { return (uint32_t)(((float)idx + 0.5f) * f.rinv); }
struct SqIx {
    int cnt, inner;
    SqDiv d;
    int o1, i1, o2, i2;
};
__device__ __forceinline__ SqIx sq_ix(int cnt, int inner, int o1, int i1, int o2, int i2) {
    SqIx x;
    x.cnt = cnt; x.inner = inner; x.d = sq_fdiv(inner); x.o1 = o1; x.i1 = i1; x.o2 = o2; x.i2 = i2;
    return x;
}
__device__ __forceinline__ void sq_off(const SqIx& x, int idx, int& off1, int& off2) {
    const int o = (int)sq_divi((uint32_t)idx, x.d), r = idx - o * x.inner;
    off1 = o * x.o1 + r * x.i1;
    off2 = o * x.o2 + r * x.i2;
}
// K table: kt[k] = {A offset, B offset}, padded to a multiple of 16 rows; rows past cnt repeat row 0 (masked)
// (non-inlined helpers take their index descriptors as scalars: a struct argument would travel through the stack,
// i.e. through scratch memory, ~1 us per call)
SQ_FN void sq_ktab_f(lds_i2* kt, int cnt, int inner, int o1, int i1, int o2, int i2) {
    const SqIx K = sq_ix(cnt, inner, o1, i1, o2, i2);
    const int kpad = (K.cnt + 15) & ~15;
    for (int k = threadIdx.x; k < kpad; k += SQ_THREADS) {
        int a, b;
        sq_off(K, k < K.cnt ? k : 0, a, b);
        kt[k] = (i32x2){a, b};
    }
}
__device__ __forceinline__ void sq_ktab(int2* kt, const SqIx& K) { sq_ktab_f(SQL2(kt), K.cnt, K.inner, K.o1, K.i1, K.o2, K.i2); }

// D(m, n) (=|+=) sum_k A(m, k) * B(k, n) [+ bias] [LeakyReLU].  All pointers are LDS.  16x16 output tiles go
// round-robin over the waves, starting at wave `wrot` (consecutive small GEMMs of a phase start on different waves
// and run side by side).  A lane: row l&15, k = l>>4; B lane: k = l>>4, column l&15; D lane: rows (l>>4)*4 + r.
// A wave keeps TWO tiles in flight (their read -> MFMA chains are independent), four k-steps per iteration, every
// LDS read of the group issued before its first MFMA: the loop is bound by LDS and MFMA latency, not by issue.
// bias index: the OUTER part of n (conv: channel; transposed conv: n = (channel, tap)), or n itself (bias_n).
enum { SQ_STORE = 0, SQ_ACC = 1 };
struct SqTile {
    const lds_f *Ap, *Bp;
    int mt, n, d_n;
    bool nvalid;
};
__device__ __forceinline__ SqTile sq_tile(const SqIx& M, const SqIx& N, int t, int ntiles, const lds_f* A, const lds_f* Bm,
                                          int li) {
    SqTile q;
    q.mt = t / ntiles;
    const int nt = t - q.mt * ntiles;
    int a_m, b_n, unused;
    sq_off(M, min(q.mt * 16 + li, M.cnt - 1), a_m, unused);
    q.n = nt * 16 + li;
    q.nvalid = q.n < N.cnt;
    sq_off(N, q.nvalid ? q.n : N.cnt - 1, b_n, q.d_n);
    q.Ap = A + a_m;
    q.Bp = Bm + b_n;
    return q;
}
__device__ __forceinline__ void sq_tile_out(const SqIx& M, const SqIx& N, const SqTile& q, const f32x4& acc, lds_f* D,
                                            const lds_f* bias, int flags, float slope, int lk) {
    const float bv = (bias != nullptr && q.nvalid) ? bias[(flags & 4) ? (uint32_t)q.n : sq_divi((uint32_t)q.n, N.d)] : 0.f;
    for (int r = 0; r < 4; ++r) {
        const int m = q.mt * 16 + lk * 4 + r;
        if (m < M.cnt && q.nvalid) {
            int d_m, unused;
            sq_off(M, m, unused, d_m);
            float v = acc[r] + bv;
            if (flags & 2) v = v > 0.f ? v : v * slope;
            if (flags & 1) D[d_m + q.d_n] += v;
            else D[d_m + q.d_n] = v;
        }
    }
}
// flags: 1 accumulate (SQ_ACC), 2 LeakyReLU, 4 bias indexed by n
SQ_FN void sq_gemm_f(int Mc, int Mn, int Mo1, int Mi1, int Mo2, int Mi2, int Nc, int Nn, int No1, int Ni1, int No2, int Ni2,
                     int Kcnt, const lds_i2* kt, const lds_f* A, const lds_f* Bm, lds_f* D, const lds_f* bias, int flags,
                     float slope, int wrot, lds_f* ksc) {
    const SqIx M = sq_ix(Mc, Mn, Mo1, Mi1, Mo2, Mi2), N = sq_ix(Nc, Nn, No1, Ni1, No2, Ni2);
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = (wid - wrot) & (SQ_WAVES - 1);
    const int mtiles = (M.cnt + 15) >> 4, ntiles = (N.cnt + 15) >> 4, ntot = mtiles * ntiles;
    const int kpad = (Kcnt + 15) & ~15;
    const lds_i2* ktl = kt + lk;
    if (ksc != nullptr && 2 * ntot <= SQ_WAVES && kpad >= 64) {
        // Few tiles, long K (a weight gradient: K = the chunk's pixels): the K range is split over ks waves per tile.
        // Every wave leaves its partial tile in LDS, and the tile's first wave adds the slices in a fixed order
        // (deterministic) and writes the result.  All waves of the workgroup call with the same arguments, so the
        // barriers are uniform.
        const int ks = SQ_WAVES / ntot;
        const int t = wid / ks, slice = wid - t * ks;
        const bool active = t < ntot;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        SqTile q = sq_tile(M, N, active ? t : 0, ntiles, A, Bm, li);
        if (active) {
            const int iters = kpad >> 4, per = (iters + ks - 1) / ks;
            const int k_lo = slice * per * 16, k_hi = min(kpad, (slice + 1) * per * 16);
            for (int k4 = k_lo; k4 < k_hi; k4 += 16) {
                const i32x2 o0 = ktl[k4], o1 = ktl[k4 + 4], o2 = ktl[k4 + 8], o3 = ktl[k4 + 12];
                float a0 = q.Ap[o0.x], a1 = q.Ap[o1.x], a2 = q.Ap[o2.x], a3 = q.Ap[o3.x];
                float b0 = q.Bp[o0.y], b1 = q.Bp[o1.y], b2 = q.Bp[o2.y], b3 = q.Bp[o3.y];
                const int kk = k4 + lk;
                const bool v0 = kk < Kcnt, v1 = kk + 4 < Kcnt, v2 = kk + 8 < Kcnt, v3 = kk + 12 < Kcnt;
                a0 = v0 ? a0 : 0.f; b0 = v0 ? b0 : 0.f;
                a1 = v1 ? a1 : 0.f; b1 = v1 ? b1 : 0.f;
                a2 = v2 ? a2 : 0.f; b2 = v2 ? b2 : 0.f;
                a3 = v3 ? a3 : 0.f; b3 = v3 ? b3 : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
            }
            ((lds_f4*)ksc)[wid * 64 + lane] = acc;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (active && slice == 0) {
            f32x4 tot = {0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < ks; ++sl) tot += ((lds_f4*)ksc)[(t * ks + sl) * 64 + lane];
            sq_tile_out(M, N, q, tot, D, bias, flags, slope, lk);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }
    for (int t = w; t < ntot; t += 2 * SQ_WAVES) {
        const bool two = t + SQ_WAVES < ntot;  // (wave-uniform)
        const SqTile q0 = sq_tile(M, N, t, ntiles, A, Bm, li);
        const SqTile q1 = sq_tile(M, N, two ? t + SQ_WAVES : t, ntiles, A, Bm, li);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int k4 = 0; k4 < kpad; k4 += 16) {
            const i32x2 o0 = ktl[k4], o1 = ktl[k4 + 4], o2 = ktl[k4 + 8], o3 = ktl[k4 + 12];
            float a0 = q0.Ap[o0.x], a1 = q0.Ap[o1.x], a2 = q0.Ap[o2.x], a3 = q0.Ap[o3.x];
            float b0 = q0.Bp[o0.y], b1 = q0.Bp[o1.y], b2 = q0.Bp[o2.y], b3 = q0.Bp[o3.y];
            float c0 = q1.Ap[o0.x], c1 = q1.Ap[o1.x], c2 = q1.Ap[o2.x], c3 = q1.Ap[o3.x];
            float e0 = q1.Bp[o0.y], e1 = q1.Bp[o1.y], e2 = q1.Bp[o2.y], e3 = q1.Bp[o3.y];
            // rows of the K table past Kcnt point at valid but possibly stale LDS words (a shorter last chunk):
            // both operands are zeroed there, since 0 * NaN would poison the accumulator
            const int kk = k4 + lk;
            const bool v0 = kk < Kcnt, v1 = kk + 4 < Kcnt, v2 = kk + 8 < Kcnt, v3 = kk + 12 < Kcnt;
            a0 = v0 ? a0 : 0.f; b0 = v0 ? b0 : 0.f; c0 = v0 ? c0 : 0.f; e0 = v0 ? e0 : 0.f;
            a1 = v1 ? a1 : 0.f; b1 = v1 ? b1 : 0.f; c1 = v1 ? c1 : 0.f; e1 = v1 ? e1 : 0.f;
            a2 = v2 ? a2 : 0.f; b2 = v2 ? b2 : 0.f; c2 = v2 ? c2 : 0.f; e2 = v2 ? e2 : 0.f;
            a3 = v3 ? a3 : 0.f; b3 = v3 ? b3 : 0.f; c3 = v3 ? c3 : 0.f; e3 = v3 ? e3 : 0.f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0, e0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, e1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2, e2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c3, e3, acc1, 0, 0, 0);
        }
        sq_tile_out(M, N, q0, acc0, D, bias, flags, slope, lk);
        if (two) sq_tile_out(M, N, q1, acc1, D, bias, flags, slope, lk);
    }
}
__device__ __forceinline__ void sq_gemm(const SqIx& M, const SqIx& N, int Kcnt, const int2* kt, const float* A,
                                        const float* Bm, float* D, const float* bias, int mode, int act, float slope,
                                        int wrot = 0, int bias_n = 0, float* ksc = nullptr) {
    sq_gemm_f(M.cnt, M.inner, M.o1, M.i1, M.o2, M.i2, N.cnt, N.inner, N.o1, N.i1, N.o2, N.i2, Kcnt, SQL2(kt), SQL(A), SQL(Bm),
              SQL(D), bias != nullptr ? SQL(bias) : (lds_f*)nullptr,
              (mode == SQ_ACC ? 1 : 0) | (act ? 2 : 0) | (bias_n ? 4 : 0), slope, wrot, ksc != nullptr ? SQL(ksc) : (lds_f*)nullptr);
}

// ---- LDS bump allocation (identical on host and device) -----------------------------------------------------
struct SqLds {
    int off;
    __host__ __device__ int take(int floats) {
        const int o = off;
        off += (floats + 3) & ~3;
        return o;
    }
};

// ---- sums over 16-lane groups (one DPP row) -----------------------------------------------------------------
// All-reduce inside every 16-lane row: xor-1 and xor-2 inside the quads, then the half-row and the row mirror.
// Four vector instructions, no LDS; fixed order => deterministic.  The result is valid in all 16 lanes.
template <int CTRL>
__device__ __forceinline__ float sq_dpp(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sq_row_sum(float v) {
    v = sq_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v = sq_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v = sq_dpp<0x141>(v);  // row_half_mirror
    v = sq_dpp<0x140>(v);  // row_mirror
    return v;
}

// ---- BatchNorm statistics ---------------------------------------------------------------------------------
__device__ __forceinline__ void sq_chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    const float tot = n + nb;
    if (tot > 0.f) {
        const float delta = mb - mean, f = nb / tot;
        mean = fmaf(delta, f, mean);
        m2 += m2b + delta * delta * n * f;
    }
    n = tot;
}
// Exact two-pass {n, mean, M2} of every channel of an LDS tile [sb][C][Lw], merged into the workgroup's running
// statistics wst[3][C] (LDS).  16 lanes per channel, 4 channels per wave at a time.
SQ_FN void sq_tile_stats_f(const lds_f* tile, int sb, int C, int Lw, lds_f* wst) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gl = lane & 15;
    const int cnt = sb * Lw;
    const SqDiv dL = sq_fdiv(Lw);
    for (int c0 = wid * 4; c0 < C; c0 += SQ_WAVES * 4) {
        const int c = c0 + (lane >> 4);
        const int cc = c < C ? c : C - 1;
        float s = 0.f;
        for (int i = gl; i < cnt; i += 16) {
            const int sidx = (int)sq_divi((uint32_t)i, dL);
            s += tile[(sidx * C + cc) * Lw + (i - sidx * Lw)];
        }
        const float mean = sq_row_sum(s) / (float)cnt;
        float q = 0.f;
        for (int i = gl; i < cnt; i += 16) {
            const int sidx = (int)sq_divi((uint32_t)i, dL);
            const float dv = tile[(sidx * C + cc) * Lw + (i - sidx * Lw)] - mean;
            q = fmaf(dv, dv, q);
        }
        q = sq_row_sum(q);
        if (gl == 0 && c < C) {
            float n0 = wst[c], m0 = wst[C + c], q0 = wst[2 * C + c];
            sq_chan_merge(n0, m0, q0, (float)cnt, mean, q);
            wst[c] = n0; wst[C + c] = m0; wst[2 * C + c] = q0;
        }
    }
}
// {sum g, sum g*xhat} of every channel of the gradient tile G [sb][C][Lw] (XH: same layout), added to wst[2][C]
SQ_FN void sq_tile_bsums_f(const lds_f* G, const lds_f* XH, int sb, int C, int Lw, lds_f* wst) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gl = lane & 15;
    const int cnt = sb * Lw;
    const SqDiv dL = sq_fdiv(Lw);
    for (int c0 = wid * 4; c0 < C; c0 += SQ_WAVES * 4) {
        const int c = c0 + (lane >> 4);
        const int cc = c < C ? c : C - 1;
        float s1 = 0.f, s2 = 0.f;
        for (int i = gl; i < cnt; i += 16) {
            const int sidx = (int)sq_divi((uint32_t)i, dL);
            const int e = (sidx * C + cc) * Lw + (i - sidx * Lw);
            const float gv = G[e];
            s1 += gv;
            s2 = fmaf(gv, XH[e], s2);
        }
        s1 = sq_row_sum(s1);
        s2 = sq_row_sum(s2);
        if (gl == 0 && c < C) { wst[c] += s1; wst[C + c] += s2; }
    }
}
// acc[c] += sum over (s, l) of tile [sb][C][Lw]   (bias gradients)
SQ_FN void sq_chan_sums_f(const lds_f* tile, int sb, int C, int Lw, lds_f* acc) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gl = lane & 15;
    const int cnt = sb * Lw;
    const SqDiv dL = sq_fdiv(Lw);
    for (int c0 = wid * 4; c0 < C; c0 += SQ_WAVES * 4) {
        const int c = c0 + (lane >> 4);
        const int cc = c < C ? c : C - 1;
        float s = 0.f;
        for (int i = gl; i < cnt; i += 16) {
            const int sidx = (int)sq_divi((uint32_t)i, dL);
            s += tile[(sidx * C + cc) * Lw + (i - sidx * Lw)];
        }
        s = sq_row_sum(s);
        if (gl == 0 && c < C) acc[c] += s;
    }
}
#define sq_tile_stats(t, sb, C, Lw, w) sq_tile_stats_f(SQL(t), sb, C, Lw, SQL(w))
#define sq_tile_bsums(G, XH, sb, C, Lw, w) sq_tile_bsums_f(SQL(G), SQL(XH), sb, C, Lw, SQL(w))
#define sq_chan_sums(t, sb, C, Lw, a) sq_chan_sums_f(SQL(t), sb, C, Lw, SQL(a))
__device__ __forceinline__ void sq_write_rows(float* rows, const float* wst, int nq, int C) {
    for (int i = threadIdx.x; i < nq * C; i += SQ_THREADS) rows[(int64_t)i * gridDim.x + blockIdx.x] = wst[i];
}

// Consumer prologue: BatchNorm coefficients of NL producer layers (both encoders of a level at once) into their
// coef[4][C] blocks (LDS): mean, rstd, gamma*rstd, beta.
// training: merges the per-workgroup rows {n_w, mean_w, M2_w} (layout [3][C][rows], rows = gridDim.x <= 256).  16
// lanes per channel; a lane keeps its <= 16 rows of every layer in registers, and the loads of ALL layers are issued
// before the first sum, so the prologue costs one memory latency, not one per layer.  Sums are taken around the
// pivot p = mean of row 0, which is within ~std/sqrt(n_w) of the batch mean: mean = p + S1/S0, M2 = S2 - S1^2/S0 with
// S1^2/S0 << S2 (no cancellation).  Workgroup 0 also stores the save block and moves the running statistics.
// eval: running statistics.  Caller synchronises afterwards.
struct SqBnRef {  // the fields of a layer its BatchNorm helpers need (kernel arguments are not addressable by a callee)
    const float *fstat, *bstat, *gamma, *beta;
    float *rmean, *rvar, *save;
    int cout, lout;
    float bn_eps, bn_mom;
    int64_t ga_off, be_off;
    const float *fgath, *bgath;  // world > 1: the gathered rows of the ranks (strides fgs / bgs floats)
    int world, rank, fgs, bgs;
};
__device__ __forceinline__ SqBnRef sq_bnref(const SqLayer& y, const float* gath, int world, int rank) {
    SqBnRef r;
    r.world = world; r.rank = rank; r.fgs = y.fg_stride; r.bgs = y.bg_stride;
    r.fgath = gath + y.fg_off; r.bgath = gath + y.bg_off;
    r.fstat = y.fstat; r.bstat = y.bstat; r.gamma = y.gamma; r.beta = y.beta; r.rmean = y.rmean; r.rvar = y.rvar;
    r.save = y.save; r.cout = y.cout; r.lout = y.lout; r.bn_eps = y.bn_eps; r.bn_mom = y.bn_mom; r.ga_off = y.ga_off;
    r.be_off = y.be_off;
    return r;
}
template <int NL>
__device__ __forceinline__ void sq_bn_prepare_f(const SqBnRef& P0, const SqBnRef& P1, lds_f* coef0, lds_f* coef1, int training) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gl = lane & 15;
    // data parallel: the rows are the ranks' merged rows (sq_rank_rows_k + all-gather), [rank][3][C]
    const bool dp = P0.world > 1;
    const int rows = dp ? P0.world : gridDim.x;
    const int Cmax = NL == 2 ? max(P0.cout, P1.cout) : P0.cout;
    for (int c0 = wid * 4; c0 < Cmax; c0 += SQ_WAVES * 4) {
        const int c = c0 + (lane >> 4);
        float vn[NL][16], vm[NL][16], vq[NL][16], pv[NL], gam[NL], bet[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const SqBnRef& P = l == 0 ? P0 : P1;
            const int cc = c < P.cout ? c : P.cout - 1;
            gam[l] = P.gamma[cc];
            bet[l] = P.beta[cc];
        }
        if (training) {
            // rows % 4 == 0 (any batch that fills the grid): 16-byte loads, a quarter of the load instructions (the
            // texture-address unit takes 16 cycles per wave instruction whatever its width)
            const bool vec = !dp && (rows & 3) == 0;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const SqBnRef& P = l == 0 ? P0 : P1;
                const int C = P.cout, cc = c < C ? c : C - 1;
                gl_cf* rn = (gl_cf*)(dp ? P.fgath + cc : P.fstat + (int64_t)cc * rows);
                gl_cf* rm = (gl_cf*)(dp ? P.fgath + C + cc : P.fstat + (int64_t)(C + cc) * rows);
                gl_cf* rq = (gl_cf*)(dp ? P.fgath + 2 * C + cc : P.fstat + (int64_t)(2 * C + cc) * rows);
                const int rs = dp ? P.fgs : 1;
                pv[l] = rm[0];
                if (vec) {
                    const int r4n = rows >> 2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (always load from a valid row and zero the count afterwards: a predicated load becomes a
                        // branch around it, and the loads then no longer issue back to back)
                        const int r4 = gl + 16 * j;
                        const bool ok = r4 < r4n;
                        const int rr = ok ? r4 : 0;
                        const f32x4 a = ((gl_cf4*)rn)[rr], b = ((gl_cf4*)rm)[rr], q = ((gl_cf4*)rq)[rr];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            vn[l][4 * j + u] = ok ? a[u] : 0.f; vm[l][4 * j + u] = b[u]; vq[l][4 * j + u] = ok ? q[u] : 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int r = gl + 16 * j;
                        const bool ok = r < rows;
                        const int rr = ok ? r * rs : 0;
                        const float a = rn[rr], b = rm[rr], q = rq[rr];
                        vn[l][j] = ok ? a : 0.f; vm[l][j] = b; vq[l][j] = ok ? q : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const SqBnRef& P = l == 0 ? P0 : P1;
            lds_f* coef = l == 0 ? coef0 : coef1;
            const int C = P.cout, cc = c < C ? c : C - 1;
            float mean, var, ntot = 0.f;
            if (training) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float dm = vm[l][j] - pv[l];
                    s0 += vn[l][j];
                    s1 = fmaf(vn[l][j], dm, s1);
                    s2 += fmaf(vn[l][j] * dm, dm, vq[l][j]);
                }
                s0 = sq_row_sum(s0); s1 = sq_row_sum(s1); s2 = sq_row_sum(s2);
                ntot = s0;
                const float dmean = s1 / s0;
                mean = pv[l] + dmean;
                var = fmaxf(s2 - s1 * dmean, 0.f) / s0;
            } else {
                mean = P.rmean[cc];
                var = P.rvar[cc];
            }
            if (gl == 0 && c < C) {
                const float rstd = rsqrtf(var + P.bn_eps);
                const float sc = gam[l] * rstd, sh = bet[l];
                coef[c] = mean; coef[C + c] = rstd; coef[2 * C + c] = sc; coef[3 * C + c] = sh;
                if (training && blockIdx.x == 0) {
                    P.save[c] = mean; P.save[C + c] = rstd; P.save[2 * C + c] = sc; P.save[3 * C + c] = sh;
                    const float unbiased = ntot > 1.f ? var * ntot / (ntot - 1.f) : var;
                    P.rmean[c] = (1.f - P.bn_mom) * P.rmean[c] + P.bn_mom * mean;
                    P.rvar[c] = (1.f - P.bn_mom) * P.rvar[c] + P.bn_mom * unbiased;
                }
            }
        }
    }
}
#define SQ_BNREF(P) sq_bnref(P, d.gath, d.world, d.rank)
#define sq_bn_prepare(P, coef, training) sq_bn_prepare_f<1>(SQ_BNREF(P), SQ_BNREF(P), SQL(coef), SQL(coef), training)
#define sq_bn_prepare2(P0, c0, P1, c1, training) sq_bn_prepare_f<2>(SQ_BNREF(P0), SQ_BNREF(P1), SQL(c0), SQL(c1), training)

// Backward prologue for the BatchNorm of NL layers: coef[4C..6C) = {m1 = sum g / n, m2 = sum g*xhat / n} from the rows
// the previous phase left (coef[0..4C), the save block, is copied by the caller's LDS-DMA).  Every workgroup puts
// its share of dgamma / dbeta into its partial row (workgroup 0: the sums, the others: zero).
template <int NL>
__device__ __forceinline__ void sq_bn_bwd_prepare_f(const SqBnRef& L0, const SqBnRef& L1, lds_f* coef0, lds_f* coef1, float* prow, int B) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gl = lane & 15;
    // data parallel: rows = the ranks' sums [rank][2C + 4] (the count of the rank's values per channel at [2C]);
    // dx needs the means over the global batch, dgamma / dbeta stay this rank's sums (the gradient all-reduce adds them)
    const bool dp = L0.world > 1;
    const int rows = dp ? L0.world : gridDim.x;
    const int Cmax = NL == 2 ? max(L0.cout, L1.cout) : L0.cout;
    for (int c0 = wid * 4; c0 < Cmax; c0 += SQ_WAVES * 4) {
        const int c = c0 + (lane >> 4);
        float v1[NL][16], v2[NL][16], vc[NL], own1[NL], own2[NL];
        const bool vec = !dp && (rows & 3) == 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const SqBnRef& Lr = l == 0 ? L0 : L1;
            const int C = Lr.cout, cc = c < C ? c : C - 1;
            gl_cf* r1 = (gl_cf*)(dp ? Lr.bgath + cc : Lr.bstat + (int64_t)cc * rows);
            gl_cf* r2 = (gl_cf*)(dp ? Lr.bgath + C + cc : Lr.bstat + (int64_t)(C + cc) * rows);
            const int rs = dp ? Lr.bgs : 1;
            vc[l] = 0.f; own1[l] = own2[l] = 0.f;
            if (dp) {
                gl_cf* rc = (gl_cf*)(Lr.bgath + 2 * C);
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // world <= 64
                    const int r = gl + 16 * j;
                    const float n = rc[r < rows ? r * rs : 0];
                    vc[l] += r < rows ? n : 0.f;
                }
                own1[l] = r1[Lr.rank * rs];
                own2[l] = r2[Lr.rank * rs];
            }
            if (vec) {
                const int r4n = rows >> 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r4 = gl + 16 * j;
                    const bool ok = r4 < r4n;
                    const int rr = ok ? r4 : 0;
                    const f32x4 a = ((gl_cf4*)r1)[rr], b = ((gl_cf4*)r2)[rr];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { v1[l][4 * j + u] = ok ? a[u] : 0.f; v2[l][4 * j + u] = ok ? b[u] : 0.f; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int r = gl + 16 * j;
                    const bool ok = r < rows;
                    const int rr = ok ? r * rs : 0;
                    const float a = r1[rr], b = r2[rr];
                    v1[l][j] = ok ? a : 0.f;
                    v2[l][j] = ok ? b : 0.f;
                }
            }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const SqBnRef& Lr = l == 0 ? L0 : L1;
            lds_f* coef = l == 0 ? coef0 : coef1;
            const int C = Lr.cout;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { s1 += v1[l][j]; s2 += v2[l][j]; }
            s1 = sq_row_sum(s1);
            s2 = sq_row_sum(s2);
            float ntot = (float)B * (float)Lr.lout;
            if (dp) ntot = sq_row_sum(vc[l]);
            if (gl == 0 && c < C) {
                const float ninv = 1.f / ntot;
                coef[4 * C + c] = s1 * ninv;
                coef[5 * C + c] = s2 * ninv;
                prow[Lr.be_off + c] = blockIdx.x == 0 ? (dp ? own1[l] : s1) : 0.f;
                prow[Lr.ga_off + c] = blockIdx.x == 0 ? (dp ? own2[l] : s2) : 0.f;
            }
        }
    }
    // parameters sit 4-float aligned in the flat buffer: the padding words of a row are zeroed too, so that
    // sq_reduce_k never sums uninitialised memory
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const SqBnRef& Lr = l == 0 ? L0 : L1;
        for (int c = Lr.cout + (int)threadIdx.x; c < ((Lr.cout + 3) & ~3); c += SQ_THREADS)
            prow[Lr.be_off + c] = prow[Lr.ga_off + c] = 0.f;
    }
}
#define sq_bn_bwd_prepare(Lr, coef, prow, B) sq_bn_bwd_prepare_f<1>(SQ_BNREF(Lr), SQ_BNREF(Lr), SQL(coef), SQL(coef), prow, B)
#define sq_bn_bwd_prepare2(L0, c0, L1, c1, prow, B) sq_bn_bwd_prepare_f<2>(SQ_BNREF(L0), SQ_BNREF(L1), SQL(c0), SQL(c1), prow, B)

// ---- staging -----------------------------------------------------------------------------------------------
// Every global load of a stage is ISSUED before the first one is waited for: contiguous copies go global -> LDS by
// LDS-DMA (global_load_lds, 16 bytes per lane, no register destination), tensors that need arithmetic on the way in
// (BatchNorm, gradient transform) are held in <= 2 float4 registers per lane (SqR2) between sq_*_issue and
// sq_*_finish.  One latency round per stage instead of one per tensor.
__device__ __forceinline__ bool sq_vec_ok(const void* p, int n) { return (((uintptr_t)p) & 15) == 0 && (n & 3) == 0; }

// dst (LDS, 16-byte aligned) <- src[0..n).  Asynchronous: sq_bar_dma() before the data is read.
SQ_FN void sq_dma_in_f(lds_f* dst, const float* __restrict__ src, int n) {
    if (sq_vec_ok(src, n)) {
        const int lane = threadIdx.x & 63;
        const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int n4 = n >> 2;
        for (int p0 = wid * 64; p0 < n4; p0 += SQ_WAVES * 64) {  // one 1-KiB piece per wave instruction
            if (p0 + lane < n4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (p0 + lane)),
                                                 (__attribute__((address_space(3))) void*)(dst + 4 * p0), 16, 0, 0);
        }
    } else {
        for (int e = threadIdx.x; e < n; e += SQ_THREADS) dst[e] = src[e];
    }
}
#define sq_dma_in(dst, src, n) sq_dma_in_f(SQL(dst), src, n)
// Barrier that orders LDS only.  __syncthreads() also waits for every outstanding GLOBAL store of the wave
// (2-3 us each time under load); the phases only ever hand data over through LDS inside a launch, and a store's data
// has left its registers when the instruction issues, so the stores may stay in flight across the barrier.
__device__ __forceinline__ void sq_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and the one that also retires the LDS-DMA copies issued before it
__device__ __forceinline__ void sq_bar_dma() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void sq_copy_out(float* __restrict__ dst, const float* src, int n) {
    if (sq_vec_ok(dst, n)) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int e = threadIdx.x; e < (n >> 2); e += SQ_THREADS) d4[e] = s4[e];
    } else {
        for (int e = threadIdx.x; e < n; e += SQ_THREADS) dst[e] = src[e];
    }
}
__device__ __forceinline__ int sq_chan_of(int e, int C, SqDiv dL, SqDiv dC) {
    const uint32_t row = sq_divi((uint32_t)e, dL);
    return (int)(row - sq_divi(row, dC) * (uint32_t)C);
}

struct SqR2 {
    float4 a, b;
    int fast;  // 1: the tensor chunk is in a / b (float4 index tid and tid + SQ_THREADS); 0: finish reads memory itself
};
__device__ __forceinline__ void sq_r2_issue(SqR2& r, const float* __restrict__ src, int n) {
    r.fast = sq_vec_ok(src, n) && n <= 8 * SQ_THREADS;
    if (r.fast) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        const int n4 = n >> 2, t = threadIdx.x;
        if (t < n4) r.a = s4[t];
        if (t + SQ_THREADS < n4) r.b = s4[t + SQ_THREADS];
    }
}

// tile [sb][C][Lw] <- BatchNorm(y) with coef (mean, rstd, scale, shift); xh (optional) <- the normalised value
__device__ __forceinline__ void sq_bn4(const float4& v4, int e4, lds_f* tile, lds_f* xh, const lds_f* coef, int C, SqDiv dL,
                                       SqDiv dC) {
    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
    float t[4], h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = sq_chan_of(4 * e4 + q, C, dL, dC);
        const float v = vv[q] - coef[c];
        t[q] = fmaf(v, coef[2 * C + c], coef[3 * C + c]);
        h[q] = v * coef[C + c];
    }
    ((lds_f4*)tile)[e4] = (f32x4){t[0], t[1], t[2], t[3]};
    if (xh != nullptr) ((lds_f4*)xh)[e4] = (f32x4){h[0], h[1], h[2], h[3]};
}
__device__ __forceinline__ void sq_stage_bn_finish_f(const SqR2& r, lds_f* tile, lds_f* xh, const float* __restrict__ y,
                                                     const lds_f* coef, int n, int C, int Lw) {
    const SqDiv dL = sq_fdiv(Lw), dC = sq_fdiv(C);
    if (r.fast) {
        const int n4 = n >> 2, t = threadIdx.x;
        if (t < n4) sq_bn4(r.a, t, tile, xh, coef, C, dL, dC);
        if (t + SQ_THREADS < n4) sq_bn4(r.b, t + SQ_THREADS, tile, xh, coef, C, dL, dC);
        return;
    }
    for (int e = threadIdx.x; e < n; e += SQ_THREADS) {
        const int c = sq_chan_of(e, C, dL, dC);
        const float v = y[e] - coef[c];
        tile[e] = fmaf(v, coef[2 * C + c], coef[3 * C + c]);
        if (xh != nullptr) xh[e] = v * coef[C + c];
    }
}
#define sq_stage_bn_finish(r, tile, xh, y, coef, n, C, Lw) \
    sq_stage_bn_finish_f(r, SQL(tile), (xh) != nullptr ? SQL(xh) : (lds_f*)nullptr, y, SQL(coef), n, C, Lw)
// dpre tile [sb][C][Lw] <- act'(y) * (has_bn ? scale*(g - m1 - xhat*m2) : g)
// sliced != 0: g is the gradient of x[:, :, 0] ([sb][C], Decoders.py:65): only position 0 of a row carries gradient
__device__ __forceinline__ float sq_dpre_of(float gv, float yv, const lds_f* coef, int has_bn, int c, int C, float slope) {
    if (has_bn) {
        const float xh = (yv - coef[c]) * coef[C + c];
        gv = coef[2 * C + c] * (gv - coef[4 * C + c] - xh * coef[5 * C + c]);
    }
    return yv > 0.f ? gv : gv * slope;
}
__device__ __forceinline__ void sq_dpre4(const float4& gv, const float4& yv, int e4, lds_f* dp, const lds_f* coef, int has_bn,
                                         int C, SqDiv dL, SqDiv dC, float slope) {
    // (opaque element index: the channels -- and with them five LDS addresses per element -- are the same in every
    // chunk, so the compiler kept them across the chunk loop of sq_bh_k, ran out of registers and spilled ten of them:
    // scratch reloads with vmcnt waits at the head of every chunk.  Recomputing them costs a few VALU operations.)
    asm volatile("" : "+v"(e4));
    f32x4 o;
    o.x = sq_dpre_of(gv.x, yv.x, coef, has_bn, has_bn ? sq_chan_of(4 * e4, C, dL, dC) : 0, C, slope);
    o.y = sq_dpre_of(gv.y, yv.y, coef, has_bn, has_bn ? sq_chan_of(4 * e4 + 1, C, dL, dC) : 0, C, slope);
    o.z = sq_dpre_of(gv.z, yv.z, coef, has_bn, has_bn ? sq_chan_of(4 * e4 + 2, C, dL, dC) : 0, C, slope);
    o.w = sq_dpre_of(gv.w, yv.w, coef, has_bn, has_bn ? sq_chan_of(4 * e4 + 3, C, dL, dC) : 0, C, slope);
    ((lds_f4*)dp)[e4] = o;
}
struct SqR2x2 {
    SqR2 g, y;
};
__device__ __forceinline__ void sq_dpre_issue(SqR2x2& r, const float* __restrict__ g, const float* __restrict__ y, int n,
                                              int sliced) {
    r.g.fast = r.y.fast = 0;
    if (sliced || !sq_vec_ok(g, n)) return;
    sq_r2_issue(r.y, y, n);
    if (r.y.fast) sq_r2_issue(r.g, g, n);
}
__device__ __forceinline__ void sq_dpre_finish_f(const SqR2x2& r, lds_f* dp, const float* __restrict__ g,
                                                 const float* __restrict__ y, const lds_f* coef, int has_bn, int n, int C, int Lw,
                                                 float slope, int sliced) {
    const SqDiv dL = sq_fdiv(Lw), dC = sq_fdiv(C);
    if (r.y.fast && r.g.fast) {
        const int n4 = n >> 2, t = threadIdx.x;
        if (t < n4) sq_dpre4(r.g.a, r.y.a, t, dp, coef, has_bn, C, dL, dC, slope);
        if (t + SQ_THREADS < n4) sq_dpre4(r.g.b, r.y.b, t + SQ_THREADS, dp, coef, has_bn, C, dL, dC, slope);
        return;
    }
    for (int e = threadIdx.x; e < n; e += SQ_THREADS) {
        const float yv = y[e];
        float gv;
        if (sliced) {
            const uint32_t row = sq_divi((uint32_t)e, dL);
            gv = (uint32_t)e == row * (uint32_t)Lw ? g[row] : 0.f;
        } else {
            gv = g[e];
        }
        dp[e] = sq_dpre_of(gv, yv, coef, has_bn, has_bn ? sq_chan_of(e, C, dL, dC) : 0, C, slope);
    }
}
#define sq_dpre_finish(r, dp, g, y, coef, has_bn, n, C, Lw, slope, sliced) \
    sq_dpre_finish_f(r, SQL(dp), g, y, SQL(coef), has_bn, n, C, Lw, slope, sliced)
__device__ __forceinline__ void sq_zero(float* p, int n) {
    for (int e = threadIdx.x; e < n; e += SQ_THREADS) p[e] = 0.f;
}

// ---- eval mode: BatchNorm folded into the consumer's weights ---------------------------------------------------
// With running statistics a BatchNorm is the per-channel affine map v -> v*sc + t (sc = gamma*rstd, t = beta - mean*sc),
// and in the 1-D models the layer behind it is linear in that value and has no padding (Conv -> LeakyReLU -> BatchNorm
// -> next Conv / Linear: Encoders.py:42-48, 84-85; multiscale_vae.py:54-63), so the map moves into the consumer's LDS
// copy of its weights once per launch: W[n][q] *= sc[c(q)], bias[n] += sum_q t[c(q)] * W[n][q] with c(q) = q / per.
// The coefficient block then becomes the identity and the staging code passes the stored activations through.
// W: rows [N][K] in LDS, complete, like bias [N] and coef (mean, rstd, scale, shift; C channels).  Ends with a barrier.
__device__ __forceinline__ void sq_fold_bn_rows(float* W, float* bias, float* coef, int N, int K, int per, int C) {
    const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    for (int n0 = 0; n0 < N; n0 += SQ_THREADS / 16) {  // 16 lanes per output row
        const int n = n0 + grp;
        float a = 0.f;
        if (n < N)
            for (int q = l16; q < K; q += 16) {
                const int c = q / per;
                a = fmaf(coef[3 * C + c] - coef[c] * coef[2 * C + c], W[n * K + q], a);
            }
        a = sq_row_sum(a);
        if (n < N && l16 == 0) bias[n] += a;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (generic pointers: flat accesses)
    for (int e = threadIdx.x; e < N * K; e += SQ_THREADS) W[e] *= coef[2 * C + (e % K) / per];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int c = threadIdx.x; c < C; c += SQ_THREADS) {
        coef[c] = 0.f; coef[C + c] = 1.f; coef[2 * C + c] = 1.f; coef[3 * C + c] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sum of v over the workgroup, valid in thread 0 (red: >= SQ_WAVES floats of LDS; fixed order)
__device__ __forceinline__ float sq_block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    sq_bar();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < SQ_WAVES; ++w) t += red[w];
    sq_bar();
    return t;
}
__device__ __forceinline__ float sq_bce_term(float p, float t) {  // F.binary_cross_entropy clamps each log at -100
    return -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f));
}
// x_rec = OUT[:, :, 0] of the last decoder layer (Decoders.py:65) and, when the program ends with the criterion, its
// reconstruction term and gradient: returns this thread's share of the reconstruction sum.
__device__ __forceinline__ float sq_xrec_out(const SqDesc& d, const SqLayer& y, const float* OUT, int s0, int sb) {
    float rec = 0.f;
    const float gs = d.has_elbo && d.egscale != nullptr ? d.egscale[0] : 1.f;
    const SqDiv dC = sq_fdiv(y.cout);
    for (int e = threadIdx.x; e < sb * y.cout; e += SQ_THREADS) {
        const float p = OUT[e * y.lout];
        const int64_t gi = (int64_t)s0 * y.cout + e;
        d.xrec[gi] = p;
        if (d.has_elbo) {
            const int sidx = (int)sq_divi((uint32_t)e, dC), c = e - sidx * y.cout;
            const float t = d.target[(int64_t)(s0 + sidx) * d.t_sb + (int64_t)c * d.t_se];
            float g;
            if (d.bce) { rec += sq_bce_term(p, t); g = (p - t) / fmaxf((1.f - p) * p, 1e-12f); }
            else { rec = fmaf(p - t, p - t, rec); g = 2.f * (p - t); }
            d.dxrec_w[gi] = g * gs;
        }
    }
    return rec;
}

// ---- forward: encoder layer i of every encoder --------------------------------------------------------------
// LDS use of the phases is computed by the same code on the host (sq_lds_*): keep the take() calls in step.
__host__ __device__ inline int sq_lds_fe(const SqDesc& d, int i, int SB) {
    SqLds l = {0};
    if (i == 0) l.take(SB * d.C * d.L);
    for (int e = 0; e < d.nenc; ++e) {
        const SqLayer& y = d.enc[e][i];
        const int K = y.cin * y.k;
        l.take(y.cout * K); l.take(y.cout); l.take(SQ_KT(K));
        if (i > 0) { l.take(4 * y.cin); l.take(SB * y.cin * y.lin); }
        l.take(SB * y.cout * y.lout); l.take(3 * y.cout);
    }
    return l.off;
}
template <int NENC>
__global__ __launch_bounds__(SQ_THREADS) void sq_fe_k(SqDesc d, int i) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SQ_TDECL;
    SqLds l = {0};
    float* X = i == 0 ? smem + l.take(d.SB * d.C * d.L) : nullptr;
    float *W[NENC], *Bi[NENC], *coef[NENC], *IN[NENC], *OUT[NENC], *wst[NENC];
    int2* kt[NENC];
    SqR2 r[NENC];
    SQ_T(0);
    int chunk = blockIdx.x;
    {   // the first chunk's loads go out before anything else: the set-up below hides their latency
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        if (i == 0) sq_dma_in(X, d.x + (int64_t)s0 * d.C * d.L, sb * d.C * d.L);
        else
            _Pragma("unroll") for (int e = 0; e < NENC; ++e)
                sq_r2_issue(r[e], d.enc[e][i - 1].y + (int64_t)s0 * d.enc[e][i].cin * d.enc[e][i].lin,
                            sb * d.enc[e][i].cin * d.enc[e][i].lin);
    }
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqLayer& y = d.enc[e][i];
        const int K = y.cin * y.k;
        W[e] = smem + l.take(y.cout * K);
        Bi[e] = smem + l.take(y.cout);
        kt[e] = reinterpret_cast<int2*>(smem + l.take(SQ_KT(K)));
        coef[e] = nullptr; IN[e] = X;
        if (i > 0) { coef[e] = smem + l.take(4 * y.cin); IN[e] = smem + l.take(d.SB * y.cin * y.lin); }
        OUT[e] = smem + l.take(d.SB * y.cout * y.lout);
        wst[e] = smem + l.take(3 * y.cout);
        sq_dma_in(W[e], y.w, y.cout * K);
        sq_dma_in(Bi[e], y.b, y.cout);
    }
    if (i > 0) {
        if (NENC == 2) sq_bn_prepare2(d.enc[0][i - 1], coef[0], d.enc[NENC - 1][i - 1], coef[NENC - 1], d.training);
        else sq_bn_prepare(d.enc[0][i - 1], coef[0], d.training);
    }
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqLayer& y = d.enc[e][i];
        sq_ktab(kt[e], sq_ix(y.cin * y.k, y.k, y.lin, 1, y.k, 1));
        sq_zero(wst[e], 3 * y.cout);
    }
    sq_bar_dma();
    if (!d.training && i > 0)  // eval: BatchNorm_{i-1} (running statistics) folded into this layer's weights
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& y = d.enc[e][i];
            sq_fold_bn_rows(W[e], Bi[e], coef[e], y.cout, y.cin * y.k, y.k, y.cin);
        }
    SQ_T(1);
    for (;;) {
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        if (i > 0)
            _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
                const SqLayer& y = d.enc[e][i];
                sq_stage_bn_finish(r[e], IN[e], nullptr, d.enc[e][i - 1].y + (int64_t)s0 * y.cin * y.lin, coef[e],
                                   sb * y.cin * y.lin, y.cin, y.lin);
            }
        sq_bar_dma();
        SQ_T(2);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& y = d.enc[e][i];
            sq_gemm(sq_ix(sb * y.lout, y.lout, y.cin * y.lin, 2, y.cout * y.lout, 1), sq_ix(y.cout, 1, y.cin * y.k, 0, y.lout, 0),
                    y.cin * y.k, kt[e], IN[e], W[e], OUT[e], Bi[e], SQ_STORE, 1, d.slope, e * (SQ_WAVES / 2));
        }
        sq_bar();
        SQ_T(3);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& y = d.enc[e][i];
            sq_copy_out(y.y + (int64_t)s0 * y.cout * y.lout, OUT[e], sb * y.cout * y.lout);
            if (d.training) sq_tile_stats(OUT[e], sb, y.cout, y.lout, wst[e]);
        }
        chunk += gridDim.x;
        if (chunk >= d.nchunks) break;
        sq_bar();
        const int n0 = chunk * d.SB, nb = min(d.SB, d.B - n0);
        if (i == 0) sq_dma_in(X, d.x + (int64_t)n0 * d.C * d.L, nb * d.C * d.L);
        else
            _Pragma("unroll") for (int e = 0; e < NENC; ++e)
                sq_r2_issue(r[e], d.enc[e][i - 1].y + (int64_t)n0 * d.enc[e][i].cin * d.enc[e][i].lin,
                            nb * d.enc[e][i].cin * d.enc[e][i].lin);
    }
    sq_bar();
    SQ_T(4);
    if (d.training)
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) sq_write_rows(d.enc[e][i].fstat, wst[e], 3, d.enc[e][i].cout);
    SQ_TPRINT("fe: setup | stage gemm out+stats:", 5);
}

// ---- forward: one transposed-conv layer on an LDS input tile ------------------------------------------------
// OUT[s][co][2t+kk] = act(sum_ci IN[s][ci][t] * W[ci][co][kk] + b[co])
__device__ __forceinline__ void sq_convT_fwd(const SqLayer& y, int sb, const int2* kt, const float* IN, const float* W,
                                             const float* Bi, float* OUT, float slope) {
    sq_gemm(sq_ix(sb * y.lin, y.lin, y.cin * y.lin, 1, y.cout * y.lout, 2), sq_ix(2 * y.cout, 2, 2, 1, y.lout, 1), y.cin, kt, IN,
            W, OUT, Bi, SQ_STORE, 1, slope);
}

// ---- forward: heads + reparameterisation + decoder Linear + convT_0 -----------------------------------------
// The mu and logvar heads of an encoder run as ONE GEMM: their weights sit back to back in LDS (n = (head, j)) and
// the result columns go to the MU / LV tiles.
__host__ __device__ inline int sq_lds_fh(const SqDesc& d, int SB) {
    SqLds l = {0};
    for (int q = 0; q < 4; ++q) l.take(SB * d.lat_total);
    for (int e = 0; e < d.nenc; ++e) {
        const SqHead& h = d.head[e];
        const SqLayer& p = d.enc[e][d.NE - 1];
        l.take(2 * d.lat * h.K); l.take(2 * d.lat);
        l.take(SQ_KT(h.K)); l.take(4 * p.cout); l.take(SB * h.K);
    }
    const SqLayer& y0 = d.dec[0];
    l.take(y0.cin * d.lat_total); l.take(y0.cin); l.take(SQ_KT(d.lat_total));
    l.take(y0.cin * y0.cout * 2); l.take(y0.cout); l.take(SQ_KT(y0.cin));
    l.take(SB * y0.cin); l.take(SB * y0.cout * y0.lout); l.take(3 * y0.cout); l.take(16);
    // eval (d.prune): the decoder at time index 0 only: per layer the folded tap-0 weights [cin][cout] and bias, and two
    // [SB][channels] vectors (always reserved: the chunk plan does not depend on the mode)
    int cmax = 1;
    for (int j = 0; j < d.ND; ++j) {
        l.take(d.dec[j].cin * d.dec[j].cout); l.take(d.dec[j].cout);
        if (d.dec[j].cout > cmax) cmax = d.dec[j].cout;
    }
    l.take(SB * cmax); l.take(SB * cmax);
    return l.off;
}
template <int NENC>
__global__ __launch_bounds__(SQ_THREADS) void sq_fh_k(SqDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SQ_TDECL;
    SqLds l = {0};
    float* MU = smem + l.take(d.SB * d.lat_total);
    float* LV = smem + l.take(d.SB * d.lat_total);
    float* EP = smem + l.take(d.SB * d.lat_total);
    float* Z = smem + l.take(d.SB * d.lat_total);
    float *WH[NENC], *bh[NENC], *coef[NENC], *IN[NENC];
    int2* kth[NENC];
    SqR2 r[NENC];
    SQ_T(0);
    int chunk = blockIdx.x;
    {
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        sq_dma_in(EP, d.eps + (int64_t)s0 * d.lat_total, sb * d.lat_total);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e)
            sq_r2_issue(r[e], d.enc[e][d.NE - 1].y + (int64_t)s0 * d.head[e].K, sb * d.head[e].K);
    }
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqHead& h = d.head[e];
        const SqLayer& p = d.enc[e][d.NE - 1];
        WH[e] = smem + l.take(2 * d.lat * h.K);
        bh[e] = smem + l.take(2 * d.lat);
        kth[e] = reinterpret_cast<int2*>(smem + l.take(SQ_KT(h.K)));
        coef[e] = smem + l.take(4 * p.cout);
        IN[e] = smem + l.take(d.SB * h.K);
        sq_dma_in(WH[e], h.wmu, d.lat * h.K); sq_dma_in(WH[e] + d.lat * h.K, h.wlv, d.lat * h.K);
        sq_dma_in(bh[e], h.bmu, d.lat); sq_dma_in(bh[e] + d.lat, h.blv, d.lat);
    }
    const SqLayer& y0 = d.dec[0];
    float* Wl = smem + l.take(y0.cin * d.lat_total);
    float* bl = smem + l.take(y0.cin);
    int2* ktl = reinterpret_cast<int2*>(smem + l.take(SQ_KT(d.lat_total)));
    float* W0 = smem + l.take(y0.cin * y0.cout * 2);
    float* b0 = smem + l.take(y0.cout);
    int2* kt0 = reinterpret_cast<int2*>(smem + l.take(SQ_KT(y0.cin)));
    float* H0 = smem + l.take(d.SB * y0.cin);
    float* OUT = smem + l.take(d.SB * y0.cout * y0.lout);
    float* wst = smem + l.take(3 * y0.cout);
    float* red = smem + l.take(16);
    // (the pruned decoder's region is walked with a running offset: an array of pointers indexed by the run-time layer
    // number would live in scratch memory)
    const int pr_base = l.off;
    int cmax = 1;
    for (int j = 0; j < d.ND; ++j) {
        l.take(d.dec[j].cin * d.dec[j].cout); l.take(d.dec[j].cout);
        cmax = max(cmax, d.dec[j].cout);
    }
    float* const PV0 = smem + l.take(d.SB * cmax);
    float* const PV1 = smem + l.take(d.SB * cmax);
    if (d.prune) {
        // Eval: x_rec = x[:, :, 0] of the last layer (Decoders.py:65; multiscale_vae.py:192), and position 0 of a
        // ConvTranspose1d(k 2, s 2) output depends only on position 0 of its input through tap 0: the decoder behind
        // the Linear layer is a chain of [cin x cout] products per sample.  BatchNorm_{j-1} (running statistics, behind
        // the activation) is folded into layer j: WF_j[ci][co] = sc[ci] * w_j[ci][co][0], bF_j[co] = b_j[co] + sum_ci
        // t[ci] * w_j[ci][co][0].
        SqLds lp = {pr_base};
        for (int j = 0; j < d.ND; ++j) {
            const SqLayer& y = d.dec[j];
            const SqLayer& pb = d.dec[j > 0 ? j - 1 : 0];
            float* const WFj = smem + lp.take(y.cin * y.cout);
            float* const bFj = smem + lp.take(y.cout);
            for (int e = threadIdx.x; e < y.cin * y.cout; e += SQ_THREADS) {
                const int ci = e / y.cout;
                float sc = 1.f;
                if (j > 0) sc = pb.gamma[ci] * rsqrtf(pb.rvar[ci] + pb.bn_eps);
                WFj[e] = sc * y.w[2 * e];
            }
            for (int co = threadIdx.x; co < y.cout; co += SQ_THREADS) {
                float a = y.b[co];
                if (j > 0)
                    for (int ci = 0; ci < y.cin; ++ci) {
                        const float sc = pb.gamma[ci] * rsqrtf(pb.rvar[ci] + pb.bn_eps);
                        a = fmaf(pb.beta[ci] - pb.rmean[ci] * sc, y.w[2 * (ci * y.cout + co)], a);
                    }
                bFj[co] = a;
            }
        }
    }
    sq_dma_in(Wl, d.wl, y0.cin * d.lat_total); sq_dma_in(bl, d.bl, y0.cin);
    sq_dma_in(W0, y0.w, y0.cin * y0.cout * 2); sq_dma_in(b0, y0.b, y0.cout);
    if (NENC == 2) sq_bn_prepare2(d.enc[0][d.NE - 1], coef[0], d.enc[NENC - 1][d.NE - 1], coef[NENC - 1], d.training);
    else sq_bn_prepare(d.enc[0][d.NE - 1], coef[0], d.training);
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) sq_ktab(kth[e], sq_ix(d.head[e].K, 1, 1, 0, 1, 0));
    sq_ktab(ktl, sq_ix(d.lat_total, 1, 1, 0, 1, 0));
    sq_ktab(kt0, sq_ix(y0.cin, 1, y0.lin, 0, 2 * y0.cout, 0));
    sq_zero(wst, 3 * y0.cout);
    sq_bar_dma();
    if (!d.training)  // eval: the encoders' last BatchNorm folded into the mu / logvar heads
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& p = d.enc[e][d.NE - 1];
            sq_fold_bn_rows(WH[e], bh[e], coef[e], 2 * d.lat, d.head[e].K, p.lout, p.cout);
        }
    SQ_T(1);
    const int lvd = (int)(LV - MU);
    float kl = 0.f, rec = 0.f;
    for (;;) {
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& p = d.enc[e][d.NE - 1];
            sq_stage_bn_finish(r[e], IN[e], nullptr, p.y + (int64_t)s0 * d.head[e].K, coef[e], sb * d.head[e].K, p.cout, p.lout);
        }
        sq_bar_dma();
        SQ_T(2);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const int K = d.head[e].K;
            // {MU | LV}[s][e*lat + j] = sum_k IN[s][k] WH[(h, j)][k] + bh[(h, j)]
            sq_gemm(sq_ix(sb, 1, K, 0, d.lat_total, 0), sq_ix(2 * d.lat, d.lat, d.lat * K, K, lvd, 1), K, kth[e], IN[e], WH[e],
                    MU + e * d.lat, bh[e], SQ_STORE, 0, 0.f, e * (SQ_WAVES / 2), 1);
        }
        sq_bar();
        SQ_T(3);
        for (int e = threadIdx.x; e < sb * d.lat_total; e += SQ_THREADS) {
            const float m = MU[e], v = LV[e];
            const float zz = fmaf(EP[e], expf(0.5f * v), m);
            Z[e] = zz;
            const int64_t gi = (int64_t)s0 * d.lat_total + e;
            d.mu[gi] = m; d.lv[gi] = v; d.z[gi] = zz;
            if (d.has_elbo) kl += 1.f + v - m * m - expf(v);
        }
        sq_bar();
        SQ_T(4);
        sq_gemm(sq_ix(sb, 1, d.lat_total, 0, y0.cin, 0), sq_ix(y0.cin, 1, d.lat_total, 0, 1, 0), d.lat_total, ktl, Z, Wl, H0, bl,
                SQ_STORE, 0, 0.f);
        sq_bar();
        SQ_T(5);
        sq_copy_out(d.h0 + (int64_t)s0 * y0.cin, H0, sb * y0.cin);
        if (d.prune) {  // the decoder at time index 0, one (sample, channel) value per thread and layer
            const float* cur = H0;
            SqLds lp = {pr_base};
            for (int j = 0; j < d.ND; ++j) {
                const SqLayer& y = d.dec[j];
                const float* const WFj = smem + lp.take(y.cin * y.cout);
                const float* const bFj = smem + lp.take(y.cout);
                float* nxt = (j & 1) ? PV1 : PV0;
                for (int e = threadIdx.x; e < sb * y.cout; e += SQ_THREADS) {
                    const int sidx = e / y.cout, co = e - sidx * y.cout;
                    float a = bFj[co];
                    for (int ci = 0; ci < y.cin; ++ci) a = fmaf(cur[sidx * y.cin + ci], WFj[ci * y.cout + co], a);
                    a = a > 0.f ? a : a * d.slope;
                    nxt[e] = a;
                    if (j == d.ND - 1) d.xrec[(int64_t)s0 * y.cout + e] = a;
                }
                cur = nxt;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        } else {
        sq_convT_fwd(y0, sb, kt0, H0, W0, b0, OUT, d.slope);
        sq_bar();
        SQ_T(6);
        sq_copy_out(y0.y + (int64_t)s0 * y0.cout * y0.lout, OUT, sb * y0.cout * y0.lout);
        if (d.ND == 1) rec += sq_xrec_out(d, y0, OUT, s0, sb);
        if (d.training && y0.has_bn) sq_tile_stats(OUT, sb, y0.cout, y0.lout, wst);
        }
        chunk += gridDim.x;
        if (chunk >= d.nchunks) break;
        sq_bar();
        const int n0 = chunk * d.SB, nb = min(d.SB, d.B - n0);
        sq_dma_in(EP, d.eps + (int64_t)n0 * d.lat_total, nb * d.lat_total);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e)
            sq_r2_issue(r[e], d.enc[e][d.NE - 1].y + (int64_t)n0 * d.head[e].K, nb * d.head[e].K);
    }
    sq_bar();
    SQ_T(7);
    if (d.training && y0.has_bn) sq_write_rows(y0.fstat, wst, 3, y0.cout);
    if (d.has_elbo) {  // this workgroup's share of the KL term (and of the reconstruction term when ND == 1)
        const float k = sq_block_sum(kl, red);
        if (threadIdx.x == 0) d.epart[2 * blockIdx.x + 1] = -0.5f * k;
        if (d.ND == 1) {
            const float rr = sq_block_sum(rec, red);
            if (threadIdx.x == 0) d.epart[2 * blockIdx.x] = rr;
        }
    }
    SQ_TPRINT("fh: setup | stage heads reparam lin convT0 out+stats:", 8);
}

// ---- forward: decoder layer j >= 1 ---------------------------------------------------------------------------
__host__ __device__ inline int sq_lds_fd(const SqDesc& d, int j, int SB) {
    const SqLayer& y = d.dec[j];
    SqLds l = {0};
    l.take(y.cin * y.cout * 2); l.take(y.cout); l.take(SQ_KT(y.cin)); l.take(4 * y.cin);
    l.take(SB * y.cin * y.lin); l.take(SB * y.cout * y.lout); l.take(3 * y.cout); l.take(16);
    return l.off;
}
__global__ __launch_bounds__(SQ_THREADS) void sq_fd_k(SqDesc d, int j) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SqLayer& y = d.dec[j];
    SqLds l = {0};
    float* W = smem + l.take(y.cin * y.cout * 2);
    float* Bi = smem + l.take(y.cout);
    int2* kt = reinterpret_cast<int2*>(smem + l.take(SQ_KT(y.cin)));
    float* coef = smem + l.take(4 * y.cin);
    float* IN = smem + l.take(d.SB * y.cin * y.lin);
    float* OUT = smem + l.take(d.SB * y.cout * y.lout);
    float* wst = smem + l.take(3 * y.cout);
    float* red = smem + l.take(16);
    SqR2 r;
    int chunk = blockIdx.x;
    sq_r2_issue(r, d.dec[j - 1].y + (int64_t)chunk * d.SB * y.cin * y.lin, min(d.SB, d.B - chunk * d.SB) * y.cin * y.lin);
    sq_dma_in(W, y.w, y.cin * y.cout * 2); sq_dma_in(Bi, y.b, y.cout);
    sq_bn_prepare(d.dec[j - 1], coef, d.training);
    sq_ktab(kt, sq_ix(y.cin, 1, y.lin, 0, 2 * y.cout, 0));
    sq_zero(wst, 3 * y.cout);
    sq_bar_dma();
    float rec = 0.f;
    for (;;) {
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        sq_stage_bn_finish(r, IN, nullptr, d.dec[j - 1].y + (int64_t)s0 * y.cin * y.lin, coef, sb * y.cin * y.lin, y.cin, y.lin);
        sq_bar();
        sq_convT_fwd(y, sb, kt, IN, W, Bi, OUT, d.slope);
        sq_bar();
        sq_copy_out(y.y + (int64_t)s0 * y.cout * y.lout, OUT, sb * y.cout * y.lout);
        if (j == d.ND - 1) rec += sq_xrec_out(d, y, OUT, s0, sb);
        if (d.training && y.has_bn) sq_tile_stats(OUT, sb, y.cout, y.lout, wst);
        chunk += gridDim.x;
        if (chunk >= d.nchunks) break;
        sq_bar();
        sq_r2_issue(r, d.dec[j - 1].y + (int64_t)chunk * d.SB * y.cin * y.lin, min(d.SB, d.B - chunk * d.SB) * y.cin * y.lin);
    }
    sq_bar();
    if (d.training && y.has_bn) sq_write_rows(y.fstat, wst, 3, y.cout);
    if (d.has_elbo && j == d.ND - 1) {
        const float rr = sq_block_sum(rec, red);
        if (threadIdx.x == 0) d.epart[2 * blockIdx.x] = rr;
    }
}

// ---- gradient accumulators of a phase: LDS blocks that leave as pieces of the workgroup's partial row ------
// The blocks of a phase are carved back to back at the END of its LDS layout (one contiguous range to zero); their
// table {LDS offset, count, flat offset} lives in LDS too (a register array indexed at run time would go to scratch).
struct SqBlocks {
    int n, first, last;
    int4* tab;  // LDS [SQ_MAXBLK]
};
// (count is rounded up to the flat buffer's 4-float parameter alignment: the padding words are zeroed with the block
// and leave with it)
__device__ __forceinline__ float* sq_blk(SqBlocks& b, float* smem, SqLds& l, int count, int64_t flat_off) {
    const int o = l.take(count);
    if (b.n == 0) b.first = o;
    b.last = l.off;
    if (threadIdx.x == 0) b.tab[b.n] = make_int4(o, (count + 3) & ~3, (int)(flat_off & 0xffffffff), (int)(flat_off >> 32));
    ++b.n;
    return smem + o;
}
__device__ __forceinline__ void sq_blk_zero(const SqBlocks& b, float* smem) {
    for (int e = (b.first >> 2) + (int)threadIdx.x; e < (b.last >> 2); e += SQ_THREADS)
        reinterpret_cast<float4*>(smem)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// (called after the chunk loop: the table entries thread 0 wrote during set-up are many barriers old)
__device__ __forceinline__ void sq_blk_flush(const SqBlocks& b, const float* smem, float* prow) {
    for (int i = 0; i < b.n; ++i) {
        const int4 t = b.tab[i];
        const int64_t flat = ((int64_t)t.w << 32) | (uint32_t)t.z;
        sq_copy_out(prow + flat, smem + t.x, t.y);
    }
}

// ---- backward: decoder layer j >= 1 --------------------------------------------------------------------------
__host__ __device__ inline int sq_lds_bd(const SqDesc& d, int j, int SB) {
    const SqLayer& y = d.dec[j];
    SqLds l = {0};
    l.take(4 * SQ_MAXBLK);
    l.take(y.cin * y.cout * 2); l.take(6 * y.cout); l.take(4 * y.cin);
    l.take(SQ_KT(2 * y.cout)); l.take(SQ_KT(SB * y.lin));
    l.take(SB * y.cin * y.lin); l.take(SB * y.cin * y.lin); l.take(SB * y.cout * y.lout); l.take(SB * y.cin * y.lin);
    l.take(2 * y.cin); l.take(SQ_WAVES * 256); l.take(y.cin * y.cout * 2); l.take(y.cout);
    return l.off;
}
// One transposed-conv layer, backward, on LDS tiles: dW[ci][(co,kk)] += sum IN*DP, db += sum DP,
// GIN[s][ci][t] = sum_(co,kk) DP[s][co][2t+kk] W[ci][co][kk]
__device__ __forceinline__ void sq_convT_bwd(const SqLayer& y, int sb, const int2* ktw, const int2* ktg, const float* IN,
                                             const float* DP, const float* W, float* GIN, float* dW, float* db, float* ksc) {
    sq_gemm(sq_ix(sb * y.lin, y.lin, y.cout * y.lout, 2, y.cin * y.lin, 1), sq_ix(y.cin, 1, 2 * y.cout, 0, y.lin, 0), 2 * y.cout, ktg,
            DP, W, GIN, nullptr, SQ_STORE, 0, 0.f, 0);
    sq_gemm(sq_ix(y.cin, 1, y.lin, 0, 2 * y.cout, 0), sq_ix(2 * y.cout, 2, y.lout, 1, 2, 1), sb * y.lin, ktw, IN, DP, dW, nullptr,
            SQ_ACC, 0, 0.f, SQ_WAVES / 2, 0, ksc);
    sq_chan_sums(DP, sb, y.cout, y.lout, db);
}
__global__ __launch_bounds__(SQ_THREADS) void sq_bd_k(SqDesc d, int j) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SqLayer& y = d.dec[j];
    const SqLayer& p = d.dec[j - 1];
    float* prow = d.part + (int64_t)blockIdx.x * d.nflat;
    // the last layer has no BatchNorm and only x[:, :, 0] leaves the decoder: its gradient is dxrec at position 0
    const int last = j == d.ND - 1;
    if (last) sq_elbo_fin(d, smem);  // (the first launch of the backward pass)
    SqLds l = {0};
    SqBlocks blk;
    blk.n = 0;
    blk.tab = reinterpret_cast<int4*>(smem + l.take(4 * SQ_MAXBLK));
    float* W = smem + l.take(y.cin * y.cout * 2);
    float* coef = smem + l.take(6 * y.cout);
    float* coefp = smem + l.take(4 * y.cin);
    int2* ktg = reinterpret_cast<int2*>(smem + l.take(SQ_KT(2 * y.cout)));
    int2* ktw = reinterpret_cast<int2*>(smem + l.take(SQ_KT(d.SB * y.lin)));
    float* IN = smem + l.take(d.SB * y.cin * y.lin);
    float* XH = smem + l.take(d.SB * y.cin * y.lin);
    float* DP = smem + l.take(d.SB * y.cout * y.lout);
    float* GIN = smem + l.take(d.SB * y.cin * y.lin);
    float* wst = smem + l.take(2 * y.cin);
    float* ksc = smem + l.take(SQ_WAVES * 256);
    float* dW = sq_blk(blk, smem, l, y.cin * y.cout * 2, y.w_off);
    float* db = sq_blk(blk, smem, l, y.cout, y.b_off);
    SqR2 ri;
    SqR2x2 rd;
    int chunk = blockIdx.x;
    auto issue = [&](int ck) {
        const int s0 = ck * d.SB, sb = min(d.SB, d.B - s0);
        sq_r2_issue(ri, p.y + (int64_t)s0 * y.cin * y.lin, sb * y.cin * y.lin);
        sq_dpre_issue(rd, last ? d.dxrec + (int64_t)s0 * y.cout : y.g + (int64_t)s0 * y.cout * y.lout,
                      y.y + (int64_t)s0 * y.cout * y.lout, sb * y.cout * y.lout, last);
    };
    issue(chunk);
    sq_dma_in(W, y.w, y.cin * y.cout * 2);
    sq_dma_in(coefp, p.save, 4 * y.cin);
    if (y.has_bn) {
        sq_dma_in(coef, y.save, 4 * y.cout);
        sq_bn_bwd_prepare(y, coef, prow, d.B);
    }
    sq_ktab(ktg, sq_ix(2 * y.cout, 2, y.lout, 1, 2, 1));
    sq_ktab(ktw, sq_ix(d.SB * y.lin, y.lin, y.cin * y.lin, 1, y.cout * y.lout, 2));
    sq_zero(wst, 2 * y.cin);
    sq_blk_zero(blk, smem);
    sq_bar_dma();
    for (;;) {
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        const float* gsrc = last ? d.dxrec + (int64_t)s0 * y.cout : y.g + (int64_t)s0 * y.cout * y.lout;
        const float* ysrc = y.y + (int64_t)s0 * y.cout * y.lout;
        sq_stage_bn_finish(ri, IN, XH, p.y + (int64_t)s0 * y.cin * y.lin, coefp, sb * y.cin * y.lin, y.cin, y.lin);
        sq_dpre_finish(rd, DP, gsrc, ysrc, coef, y.has_bn, sb * y.cout * y.lout, y.cout, y.lout, d.slope, last);
        sq_bar();
        sq_convT_bwd(y, sb, ktw, ktg, IN, DP, W, GIN, dW, db, ksc);
        sq_bar();
        sq_copy_out(p.g + (int64_t)s0 * y.cin * y.lin, GIN, sb * y.cin * y.lin);
        sq_tile_bsums(GIN, XH, sb, y.cin, y.lin, wst);
        chunk += gridDim.x;
        if (chunk >= d.nchunks) break;
        sq_bar();
        issue(chunk);
    }
    sq_bar();
    sq_write_rows(p.bstat, wst, 2, y.cin);
    sq_blk_flush(blk, smem, prow);
}

// ---- backward: convT_0 + decoder Linear + reparameterisation + heads ----------------------------------------
__host__ __device__ inline int sq_lds_bh(const SqDesc& d, int SB) {
    const SqLayer& y0 = d.dec[0];
    SqLds l = {0};
    l.take(4 * SQ_MAXBLK);
    l.take(y0.cin * y0.cout * 2); l.take(6 * y0.cout);
    l.take(SQ_KT(2 * y0.cout)); l.take(SQ_KT(SB));
    l.take(SB * y0.cin); l.take(SB * y0.cout * y0.lout); l.take(SB * y0.cin);
    l.take(y0.cin * y0.cout * 2); l.take(y0.cout);
    l.take(y0.cin * d.lat_total); l.take(SQ_KT(y0.cin)); l.take(SQ_KT(SB));
    for (int q = 0; q < 6; ++q) l.take(SB * d.lat_total);
    l.take(y0.cin * d.lat_total); l.take(y0.cin);
    for (int e = 0; e < d.nenc; ++e) {
        const SqHead& h = d.head[e];
        const SqLayer& p = d.enc[e][d.NE - 1];
        l.take(2 * d.lat * h.K); l.take(4 * p.cout); l.take(SQ_KT(SB)); l.take(SQ_KT(2 * d.lat));
        l.take(SB * h.K); l.take(SB * h.K); l.take(SB * h.K); l.take(2 * p.cout);
        l.take(d.lat * h.K); l.take(d.lat * h.K); l.take(d.lat); l.take(d.lat);
    }
    return l.off;
}
template <int NENC>
__global__ __launch_bounds__(SQ_THREADS) void sq_bh_k(SqDesc d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SQ_TDECL;
    const SqLayer& y0 = d.dec[0];
    float* prow = d.part + (int64_t)blockIdx.x * d.nflat;
    const int last = d.ND == 1;
    if (last) sq_elbo_fin(d, smem);  // (a one-layer decoder: this is the first launch of the backward pass)
    SqLds l = {0};
    SqBlocks blk;
    blk.n = 0;
    blk.tab = reinterpret_cast<int4*>(smem + l.take(4 * SQ_MAXBLK));
    SQ_T(0);
    // decoder layer 0
    float* W0 = smem + l.take(y0.cin * y0.cout * 2);
    float* coef = smem + l.take(6 * y0.cout);
    int2* ktg = reinterpret_cast<int2*>(smem + l.take(SQ_KT(2 * y0.cout)));
    int2* ktw = reinterpret_cast<int2*>(smem + l.take(SQ_KT(d.SB)));
    float* H0 = smem + l.take(d.SB * y0.cin);
    float* DP = smem + l.take(d.SB * y0.cout * y0.lout);
    float* DH = smem + l.take(d.SB * y0.cin);
    // decoder Linear + reparameterisation
    float* Wl = smem + l.take(y0.cin * d.lat_total);
    int2* ktn = reinterpret_cast<int2*>(smem + l.take(SQ_KT(y0.cin)));   // K = n (dz)
    int2* kts = reinterpret_cast<int2*>(smem + l.take(SQ_KT(d.SB)));     // K = s (dWl)
    float* Z = smem + l.take(d.SB * d.lat_total);
    float* DMU = smem + l.take(d.SB * d.lat_total);
    float* DLV = smem + l.take(d.SB * d.lat_total);
    float* MUt = smem + l.take(d.SB * d.lat_total);   // mu, logvar, eps of the chunk (reparameterisation backward)
    float* LVt = smem + l.take(d.SB * d.lat_total);
    float* EPt = smem + l.take(d.SB * d.lat_total);
    float *WH[NENC], *coefp[NENC], *IN[NENC], *XH[NENC], *GF[NENC], *wst[NENC];
    float *dWmu[NENC], *dWlv[NENC], *dbmu[NENC], *dblv[NENC];
    int2 *ktsh[NENC], *ktlat[NENC];
    SqR2 ri[NENC];
    SqR2x2 rd;
    const int dvd = (int)(DLV - DMU);
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqHead& h = d.head[e];
        const SqLayer& p = d.enc[e][d.NE - 1];
        WH[e] = smem + l.take(2 * d.lat * h.K);
        coefp[e] = smem + l.take(4 * p.cout);
        ktsh[e] = reinterpret_cast<int2*>(smem + l.take(SQ_KT(d.SB)));
        ktlat[e] = reinterpret_cast<int2*>(smem + l.take(SQ_KT(2 * d.lat)));
        IN[e] = smem + l.take(d.SB * h.K); XH[e] = smem + l.take(d.SB * h.K); GF[e] = smem + l.take(d.SB * h.K);
        wst[e] = smem + l.take(2 * p.cout);
    }
    int chunk = blockIdx.x;
    auto issue = [&](int ck) {
        const int s0 = ck * d.SB, sb = min(d.SB, d.B - s0);
        const int nl = sb * d.lat_total;
        sq_dma_in(H0, d.h0 + (int64_t)s0 * y0.cin, sb * y0.cin);
        sq_dma_in(Z, d.z + (int64_t)s0 * d.lat_total, nl);
        sq_dma_in(MUt, d.mu + (int64_t)s0 * d.lat_total, nl);
        sq_dma_in(LVt, d.lv + (int64_t)s0 * d.lat_total, nl);
        sq_dma_in(EPt, d.eps + (int64_t)s0 * d.lat_total, nl);
        sq_dpre_issue(rd, last ? d.dxrec + (int64_t)s0 * y0.cout : y0.g + (int64_t)s0 * y0.cout * y0.lout,
                      y0.y + (int64_t)s0 * y0.cout * y0.lout, sb * y0.cout * y0.lout, last);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e)
            sq_r2_issue(ri[e], d.enc[e][d.NE - 1].y + (int64_t)s0 * d.head[e].K, sb * d.head[e].K);
    };
    issue(chunk);
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqHead& h = d.head[e];
        sq_dma_in(WH[e], h.wmu, d.lat * h.K); sq_dma_in(WH[e] + d.lat * h.K, h.wlv, d.lat * h.K);
        sq_dma_in(coefp[e], d.enc[e][d.NE - 1].save, 4 * d.enc[e][d.NE - 1].cout);
    }
    sq_dma_in(W0, y0.w, y0.cin * y0.cout * 2);
    sq_dma_in(Wl, d.wl, y0.cin * d.lat_total);
    if (y0.has_bn) {
        sq_dma_in(coef, y0.save, 4 * y0.cout);
        sq_bn_bwd_prepare(y0, coef, prow, d.B);
    }
    // gradient accumulators: one contiguous range at the end of the layout
    float* dW0 = sq_blk(blk, smem, l, y0.cin * y0.cout * 2, y0.w_off);
    float* db0 = sq_blk(blk, smem, l, y0.cout, y0.b_off);
    float* dWl = sq_blk(blk, smem, l, y0.cin * d.lat_total, d.wl_off);
    float* dbl = sq_blk(blk, smem, l, y0.cin, d.bl_off);
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqHead& h = d.head[e];
        dWmu[e] = sq_blk(blk, smem, l, d.lat * h.K, h.wmu_off);
        dWlv[e] = sq_blk(blk, smem, l, d.lat * h.K, h.wlv_off);
        dbmu[e] = sq_blk(blk, smem, l, d.lat, h.bmu_off);
        dblv[e] = sq_blk(blk, smem, l, d.lat, h.blv_off);
    }
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        sq_ktab(ktsh[e], sq_ix(d.SB, 1, d.lat_total, 0, d.head[e].K, 0));
        // K = (head, n): A = {DMU | DLV}[s][e*lat + n], B = WH[(head, n)][k]
        sq_ktab(ktlat[e], sq_ix(2 * d.lat, d.lat, dvd, 1, d.lat * d.head[e].K, d.head[e].K));
        sq_zero(wst[e], 2 * d.enc[e][d.NE - 1].cout);
    }
    sq_ktab(ktg, sq_ix(2 * y0.cout, 2, y0.lout, 1, 2, 1));
    sq_ktab(ktw, sq_ix(d.SB * y0.lin, y0.lin, y0.cin * y0.lin, 1, y0.cout * y0.lout, 2));
    sq_ktab(ktn, sq_ix(y0.cin, 1, 1, 0, d.lat_total, 0));
    sq_ktab(kts, sq_ix(d.SB, 1, y0.cin, 0, d.lat_total, 0));
    sq_blk_zero(blk, smem);
    sq_bar_dma();
    SQ_T(1);
    const float gs = d.kl_grad ? (d.gscale != nullptr ? d.gscale[0] : 1.f) : 0.f;
    for (;;) {
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        const int nl = sb * d.lat_total;
        const float* gsrc = last ? d.dxrec + (int64_t)s0 * y0.cout : y0.g + (int64_t)s0 * y0.cout * y0.lout;
        const float* ysrc = y0.y + (int64_t)s0 * y0.cout * y0.lout;
        sq_dpre_finish(rd, DP, gsrc, ysrc, coef, y0.has_bn, sb * y0.cout * y0.lout, y0.cout, y0.lout, d.slope, last);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& p = d.enc[e][d.NE - 1];
            sq_stage_bn_finish(ri[e], IN[e], XH[e], p.y + (int64_t)s0 * d.head[e].K, coefp[e], sb * d.head[e].K, p.cout, p.lout);
        }
        sq_bar_dma();
        SQ_T(2);
        sq_convT_bwd(y0, sb, ktw, ktg, H0, DP, W0, DH, dW0, db0, nullptr);  // K = sb*lin is short here: no split
        sq_bar();
        SQ_T(3);
        // decoder Linear: dWl[n][k] += sum_s DH[s][n] Z[s][k];  dbl[n] += sum_s DH[s][n];  dz[s][k] = sum_n DH[s][n] Wl[n][k]
        sq_gemm(sq_ix(sb, 1, y0.cin, 0, d.lat_total, 0), sq_ix(d.lat_total, 1, 1, 0, 1, 0), y0.cin, ktn, DH, Wl, DMU, nullptr, SQ_STORE, 0,
                0.f, 0);
        sq_gemm(sq_ix(y0.cin, 1, 1, 0, d.lat_total, 0), sq_ix(d.lat_total, 1, 1, 0, 1, 0), sb, kts, DH, Z, dWl, nullptr, SQ_ACC, 0, 0.f,
                SQ_WAVES / 2);
        sq_chan_sums(DH, sb, y0.cin, 1, dbl);
        sq_bar();
        SQ_T(4);
        // reparameterisation backward (vae.py:15-18) + analytic KL gradient or the incoming mu / logvar gradients
        for (int e = threadIdx.x; e < nl; e += SQ_THREADS) {
            const int64_t gi = (int64_t)s0 * d.lat_total + e;
            const float dz = DMU[e];
            const float sd = expf(0.5f * LVt[e]);
            float a = dz, b = dz * EPt[e] * 0.5f * sd;
            if (d.gmu != nullptr) { a += d.gmu[gi]; b += d.glv[gi]; }
            if (d.kl_grad) {
                a = fmaf(gs, MUt[e], a);
                b = fmaf(gs, 0.5f * (sd * sd - 1.f), b);
            }
            DMU[e] = a;
            DLV[e] = b;
        }
        sq_bar();
        SQ_T(5);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const int K = d.head[e].K;
            // GF[s][k] = sum_(head, n) {DMU | DLV}[s][e*lat+n] WH[(head, n)][k]
            sq_gemm(sq_ix(sb, 1, d.lat_total, 0, K, 0), sq_ix(K, 1, 1, 0, 1, 0), 2 * d.lat, ktlat[e], DMU + e * d.lat, WH[e], GF[e], nullptr,
                    SQ_STORE, 0, 0.f, e * (SQ_WAVES / 2));
            // d{Wmu | Wlv}[n][k] += sum_s {DMU | DLV}[s][e*lat+n] IN[s][k]: M = (head, n)
            sq_gemm(sq_ix(2 * d.lat, d.lat, dvd, 1, (int)(dWlv[e] - dWmu[e]), K), sq_ix(K, 1, 1, 0, 1, 0), sb, ktsh[e], DMU + e * d.lat,
                    IN[e], dWmu[e], nullptr, SQ_ACC, 0, 0.f, e * (SQ_WAVES / 2) + 2);
            // bias gradients of the heads: column sums of DMU / DLV over the chunk's samples
            for (int n = threadIdx.x; n < d.lat; n += SQ_THREADS) {
                float a = 0.f, b = 0.f;
                for (int s = 0; s < sb; ++s) { a += DMU[s * d.lat_total + e * d.lat + n]; b += DLV[s * d.lat_total + e * d.lat + n]; }
                dbmu[e][n] += a;
                dblv[e][n] += b;
            }
        }
        sq_bar();
        SQ_T(6);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& p = d.enc[e][d.NE - 1];
            const int K = d.head[e].K;
            sq_copy_out(p.g + (int64_t)s0 * K, GF[e], sb * K);
            sq_tile_bsums(GF[e], XH[e], sb, p.cout, p.lout, wst[e]);
        }
        chunk += gridDim.x;
        if (chunk >= d.nchunks) break;
        sq_bar();
        issue(chunk);
    }
    sq_bar();
    SQ_T(7);
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) sq_write_rows(d.enc[e][d.NE - 1].bstat, wst[e], 2, d.enc[e][d.NE - 1].cout);
    sq_blk_flush(blk, smem, prow);
    SQ_TPRINT("bh: setup | stage convT0bwd lin reparam heads out+bsums:", 8);
}

// ---- backward: encoder layer i of every encoder -------------------------------------------------------------
__host__ __device__ inline int sq_lds_be(const SqDesc& d, int i, int SB) {
    SqLds l = {0};
    l.take(4 * SQ_MAXBLK); l.take(SQ_WAVES * 256);
    if (i == 0) l.take(SB * d.C * d.L);
    for (int e = 0; e < d.nenc; ++e) {
        const SqLayer& y = d.enc[e][i];
        const int K = y.cin * y.k;
        l.take(y.cout * K); l.take(6 * y.cout); l.take(SQ_KT(SB * y.lout));
        l.take(SB * y.cout * y.lout); l.take(y.cout * K); l.take(y.cout);
        if (i > 0) {
            l.take(4 * y.cin); l.take(SQ_KT(y.cout));
            l.take(SB * y.cin * y.lin); l.take(SB * y.cin * y.lin); l.take(SB * y.cin * y.lin); l.take(2 * y.cin);
        }
    }
    return l.off;
}
template <int NENC>
__global__ __launch_bounds__(SQ_THREADS) void sq_be_k(SqDesc d, int i) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SQ_TDECL;
    float* prow = d.part + (int64_t)blockIdx.x * d.nflat;
    SqLds l = {0};
    SqBlocks blk;
    blk.n = 0;
    blk.tab = reinterpret_cast<int4*>(smem + l.take(4 * SQ_MAXBLK));
    float* ksc = smem + l.take(SQ_WAVES * 256);
    SQ_T(0);
    float* X = i == 0 ? smem + l.take(d.SB * d.C * d.L) : nullptr;
    float *W[NENC], *coef[NENC], *coefp[NENC], *DP[NENC], *IN[NENC], *XH[NENC], *GIN[NENC];
    float *wst[NENC], *dW[NENC], *db[NENC];
    int2 *ktw[NENC], *ktg[NENC];
    SqR2 ri[NENC];
    SqR2x2 rd[NENC];
    int chunk = blockIdx.x;
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqLayer& y = d.enc[e][i];
        const int K = y.cin * y.k;
        W[e] = smem + l.take(y.cout * K);
        coef[e] = smem + l.take(6 * y.cout);
        ktw[e] = reinterpret_cast<int2*>(smem + l.take(SQ_KT(d.SB * y.lout)));
        DP[e] = smem + l.take(d.SB * y.cout * y.lout);
        coefp[e] = nullptr; ktg[e] = nullptr; IN[e] = X; XH[e] = nullptr; GIN[e] = nullptr; wst[e] = nullptr;
        if (i > 0) {
            coefp[e] = smem + l.take(4 * y.cin);
            ktg[e] = reinterpret_cast<int2*>(smem + l.take(SQ_KT(y.cout)));
            IN[e] = smem + l.take(d.SB * y.cin * y.lin);
            XH[e] = smem + l.take(d.SB * y.cin * y.lin);
            GIN[e] = smem + l.take(d.SB * y.cin * y.lin);
            wst[e] = smem + l.take(2 * y.cin);
        }
    }
    auto issue = [&](int ck) {
        const int s0 = ck * d.SB, sb = min(d.SB, d.B - s0);
        if (i == 0) sq_dma_in(X, d.x + (int64_t)s0 * d.C * d.L, sb * d.C * d.L);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& y = d.enc[e][i];
            if (i > 0) sq_r2_issue(ri[e], d.enc[e][i - 1].y + (int64_t)s0 * y.cin * y.lin, sb * y.cin * y.lin);
            sq_dpre_issue(rd[e], y.g + (int64_t)s0 * y.cout * y.lout, y.y + (int64_t)s0 * y.cout * y.lout, sb * y.cout * y.lout, 0);
        }
    };
    issue(chunk);
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqLayer& y = d.enc[e][i];
        if (i > 0) sq_dma_in(coefp[e], d.enc[e][i - 1].save, 4 * y.cin);
        sq_dma_in(W[e], y.w, y.cout * y.cin * y.k);
        sq_dma_in(coef[e], y.save, 4 * y.cout);
    }
    if (NENC == 2) sq_bn_bwd_prepare2(d.enc[0][i], coef[0], d.enc[NENC - 1][i], coef[NENC - 1], prow, d.B);
    else sq_bn_bwd_prepare(d.enc[0][i], coef[0], prow, d.B);
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {  // gradient accumulators: contiguous, at the end of the layout
        const SqLayer& y = d.enc[e][i];
        dW[e] = sq_blk(blk, smem, l, y.cout * y.cin * y.k, y.w_off);
        db[e] = sq_blk(blk, smem, l, y.cout, y.b_off);
    }
    _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
        const SqLayer& y = d.enc[e][i];
        if (i > 0) {
            sq_ktab(ktg[e], sq_ix(y.cout, 1, y.lout, 0, y.cin * 2, 0));
            sq_zero(wst[e], 2 * y.cin);
        }
        // weight gradient: K = (s, t): A = DP at s*cout*lout + t, B = IN at s*cin*lin + 2t
        sq_ktab(ktw[e], sq_ix(d.SB * y.lout, y.lout, y.cout * y.lout, 1, y.cin * y.lin, 2));
    }
    sq_blk_zero(blk, smem);
    sq_bar_dma();
    SQ_T(1);
    for (;;) {
        const int s0 = chunk * d.SB, sb = min(d.SB, d.B - s0);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& y = d.enc[e][i];
            if (i > 0) {
                sq_stage_bn_finish(ri[e], IN[e], XH[e], d.enc[e][i - 1].y + (int64_t)s0 * y.cin * y.lin, coefp[e],
                                   sb * y.cin * y.lin, y.cin, y.lin);
                sq_zero(GIN[e], sb * y.cin * y.lin);  // positions past 2*lout (odd lin) get no gradient
            }
            sq_dpre_finish(rd[e], DP[e], y.g + (int64_t)s0 * y.cout * y.lout, y.y + (int64_t)s0 * y.cout * y.lout, coef[e], 1,
                           sb * y.cout * y.lout, y.cout, y.lout, d.slope, 0);
        }
        sq_bar_dma();
        SQ_T(2);
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
            const SqLayer& y = d.enc[e][i];
            const int K = y.cin * y.k;
            // GIN[s][ci][2t+kk] = sum_co DP[s][co][t] W[co][ci][kk]   (k = 2 beyond the first layer)
            if (i > 0)
                sq_gemm(sq_ix(sb * y.lout, y.lout, y.cout * y.lout, 1, y.cin * y.lin, 2), sq_ix(2 * y.cin, 2, 2, 1, y.lin, 1), y.cout,
                        ktg[e], DP[e], W[e], GIN[e], nullptr, SQ_STORE, 0, 0.f, e * (SQ_WAVES / 2));
            // dW[co][(ci,kk)] += sum_(s,t) DP[s][co][t] IN[s][ci][2t+kk]
            sq_gemm(sq_ix(y.cout, 1, y.lout, 0, K, 0), sq_ix(K, y.k, y.lin, 1, y.k, 1), sb * y.lout, ktw[e], DP[e], IN[e], dW[e],
                    nullptr, SQ_ACC, 0, 0.f, e * (SQ_WAVES / 2) + 2, 0, ksc);
            sq_chan_sums(DP[e], sb, y.cout, y.lout, db[e]);
        }
        sq_bar();
        SQ_T(3);
        if (i > 0)
            _Pragma("unroll") for (int e = 0; e < NENC; ++e) {
                const SqLayer& y = d.enc[e][i];
                sq_copy_out(d.enc[e][i - 1].g + (int64_t)s0 * y.cin * y.lin, GIN[e], sb * y.cin * y.lin);
                sq_tile_bsums(GIN[e], XH[e], sb, y.cin, y.lin, wst[e]);
            }
        chunk += gridDim.x;
        if (chunk >= d.nchunks) break;
        sq_bar();
        issue(chunk);
    }
    sq_bar();
    SQ_T(4);
    if (i > 0)
        _Pragma("unroll") for (int e = 0; e < NENC; ++e) sq_write_rows(d.enc[e][i - 1].bstat, wst[e], 2, d.enc[e][i].cin);
    sq_blk_flush(blk, smem, prow);
    SQ_TPRINT("be: setup | stage gemms out+bsums:", 5);
}

// ---- data parallel: this rank's statistics rows -> ONE row (before the all-gather between two phases) -----------
// forward (nq = 3): Chan merge of the workgroups' {n, mean, M2} rows around the pivot mean(row 0), as sq_bn_prepare_f
// does; backward (nq = 2): plain sums of {sum g, sum g*xhat} and the rank's count of values per channel behind them.
// One wave per channel, lane l takes the rows l, l + 64, ...; the wave sum runs in a fixed order.
struct SqRankRows {
    const float* stat[SQ_MAXENC];  // [nq][C][rows]
    float* out[SQ_MAXENC];         // forward [3][C], backward [2][C] + {n}
    int C[SQ_MAXENC];
    float n_local[SQ_MAXENC];
    int rows, nq;
};
__device__ __forceinline__ float sq_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void sq_rank_rows_k(SqRankRows a) {
    const int l = blockIdx.y, lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float* stat = l == 0 ? a.stat[0] : a.stat[SQ_MAXENC - 1];
    float* out = l == 0 ? a.out[0] : a.out[SQ_MAXENC - 1];
    const int C = l == 0 ? a.C[0] : a.C[SQ_MAXENC - 1], rows = a.rows;
    if (c >= C) return;  // (wave-uniform)
    if (a.nq == 3) {
        const float pivot = stat[(int64_t)(C + c) * rows];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int r = lane; r < rows; r += 64) {
            const float n = stat[(int64_t)c * rows + r], dm = stat[(int64_t)(C + c) * rows + r] - pivot;
            const float q = stat[(int64_t)(2 * C + c) * rows + r];
            s0 += n;
            s1 = fmaf(n, dm, s1);
            s2 += fmaf(n * dm, dm, q);
        }
        s0 = sq_wave_sum(s0); s1 = sq_wave_sum(s1); s2 = sq_wave_sum(s2);
        if (lane == 0) {
            const float dmean = s1 / s0;
            out[c] = s0;
            out[C + c] = pivot + dmean;
            out[2 * C + c] = fmaxf(s2 - s1 * dmean, 0.f);
        }
    } else {
        float s1 = 0.f, s2 = 0.f;
        for (int r = lane; r < rows; r += 64) {
            s1 += stat[(int64_t)c * rows + r];
            s2 += stat[(int64_t)(C + c) * rows + r];
        }
        s1 = sq_wave_sum(s1); s2 = sq_wave_sum(s2);
        if (lane == 0) {
            out[c] = s1;
            out[C + c] = s2;
            if (c == 0) out[2 * C] = l == 0 ? a.n_local[0] : a.n_local[SQ_MAXENC - 1];
        }
    }
}

// ---- gradient rows -> flat GRAD buffer -----------------------------------------------------------------------
// grad[j] = sum over the workgroup rows, fixed order (8 slices striding over the rows, then an 8-term sum).  A
// workgroup covers SQR_COLS = 32 columns: a wave instruction then reads whole 128-byte lines of two rows (with 16
// columns every load left half of its line unused and the launch fetched 2.5 x the rows' bytes).
#define SQR_COLS 32
#define SQR_SLICES (256 / SQR_COLS)
__global__ __launch_bounds__(256) void sq_reduce_k(const float* __restrict__ part, float* __restrict__ grad, int64_t nflat,
                                                   int rows) {
    __shared__ float red[SQR_SLICES][SQR_COLS + 1];
    const int jj = threadIdx.x & (SQR_COLS - 1), slice = threadIdx.x / SQR_COLS;
    const int64_t j = (int64_t)blockIdx.x * SQR_COLS + jj;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (j < nflat) {
        int b = slice;
        for (; b + 15 * SQR_SLICES < rows; b += 16 * SQR_SLICES) {  // sixteen rows in one latency round
            float x[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = part[(int64_t)(b + SQR_SLICES * q) * nflat + j];
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                t0 += x[q];
                t1 += x[q + 1];
                t2 += x[q + 2];
                t3 += x[q + 3];
            }
        }
        for (; b + 3 * SQR_SLICES < rows; b += 4 * SQR_SLICES) {
            t0 += part[(int64_t)b * nflat + j];
            t1 += part[(int64_t)(b + SQR_SLICES) * nflat + j];
            t2 += part[(int64_t)(b + 2 * SQR_SLICES) * nflat + j];
            t3 += part[(int64_t)(b + 3 * SQR_SLICES) * nflat + j];
        }
        for (; b < rows; b += SQR_SLICES) t0 += part[(int64_t)b * nflat + j];
    }
    red[slice][jj] = (t0 + t1) + (t2 + t3);
    sq_bar();
    if (slice != 0 || j >= nflat) return;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SQR_SLICES; ++k) sum += red[k][jj];
    grad[j] = sum;
}
