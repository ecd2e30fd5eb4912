// fc.h - a run of Linear / reparameterisation ops as ONE launch.
//
// Between the last conv of the encoder and the first transposed conv of the decoder the train step is a chain of small
// row-independent ops: the two 288 -> 100 branches, the mu / logvar heads, z = mu + eps * exp(logvar / 2), and the
// decoder's two Linear layers (vae-checkpoint.ipynb:52-69, :78-84, :118-124); backward walks the same chain the other
// way.  As one launch per op that is 6 + 6 launches of 4.5 - 17 us each (a graph node costs >= 4.5 us however little it
// does), a sixth of the whole step, for 0.8 GFLOP.  fc_chain_k runs the whole run in one launch: a workgroup owns 16
// rows of the batch (one MFMA m-tile) and walks the run level by level (a level = the stages whose inputs are ready
// together); activations pass from stage to stage through LDS and are also written to global memory once, for the
// backward pass and the weight gradients (lin_wgrad_k reads them as before).
//
//   GEMM stage   out[16][N] = act( in[16][K] * W'(K,N) + bias ),  16x16x4 fp32 MFMA.  A wave owns whole 16-column
//                output tiles (the tiles of all stages of a level are dealt round-robin over the waves; the host
//                writes every wave's list of work items: tiles, element-wise stages, barriers).  A comes from LDS as
//                one ds_read_b128 per four MFMAs (lane (row, q) holds in[row][16 s + 4 q + 0..3], the k permutation
//                the B side matches); B comes straight from global memory (L2-resident weights): lane (col, q) loads
//                W'(16 s + 4 q + 0..3, n0 + col), one dwordx4 where the reduction index is contiguous (forward), four
//                coalesced dwords otherwise (input gradient).  A wave walks its tiles as ONE stream of chunks of nine
//                k-steps: the loads of chunk i+1 (the next chunk of the tile, or the first chunk of the wave's next
//                tile with that tile's bias and act'(Y) values, also when it belongs to a later level) are issued
//                before the MFMAs of chunk i, so only the very first round trip of a wave is exposed; weights do
//                not depend on activations, so this prefetch crosses the level barriers (which order LDS only).
//                Backward: the stage that produces a gradient buffer writes it raw to global memory (the weight
//                gradient wants it raw) and times act'(Y) of its consumer into LDS.
//   LOAD stage   rows of a buffer produced outside the run -> LDS (times act'(Y) in backward).
//   REPARAM      forward and backward (elem.h: reparam_fwd_k / reparam_bwd_k), element-wise on the LDS rows.
#pragma once
#include "common.h"
#include "igemm.h"  // f32x4

#define FC_ROWS 16
#ifndef FC_WAVES
#define FC_WAVES 16
#endif
#define FC_THREADS (FC_WAVES * 64)
#define FC_MAX_STAGES 12
#define FC_MAX_ITEMS 24
#ifndef FC_U
#define FC_U 9  // k-steps (of 16) per chunk (a multiple of 3: the A rows are read three k-steps at a time)
#endif

enum { FC_LOAD = 0, FC_GEMM = 1, FC_REPARAM_FWD = 2, FC_REPARAM_BWD = 3 };

struct FcStage {
    int kind, level;
    // input rows in LDS (float offsets; row strides are multiples of 4, buffers zero-padded to a multiple of 16 columns)
    int in_off, in_stride;    // GEMM: A rows;  REPARAM_FWD: mu;  REPARAM_BWD: dz
    int in2_off, in2_stride;  // GEMM: second K segment (-1: none);  REPARAM_FWD: logvar
    int K, N;                 // GEMM: reduction extent per segment, output columns;  LOAD / REPARAM: N = columns
    const float* W;           // GEMM: W'(k, n) at W[k * wsk + n * wsn]
    const float* W2;          //       second segment's weights (same strides)
    int64_t wsk, wsn;
    int vec;                  // GEMM: wsk == 1, 16-byte aligned rows: dwordx4 weight loads
    const float* bias;        // GEMM: [N] or null
    int act;                  // GEMM: output activation
    float slope;
    // outputs
    float* out;               // GEMM / REPARAM_FWD (z) / REPARAM_BWD (dmu): global rows, stride ldo
    int ldo, accumulate;
    int out_off, out_stride;  // LDS copy (-1: none).  LOAD: destination
    float* out2;              // REPARAM_BWD: dlogvar (global, stride ldo);  its LDS copy: out2_off
    int out2_off, out2_stride;
    // LDS copy of a gradient = raw * act'(Y) of the consuming op (GEMM, LOAD)
    const float* mulY;
    int has_bias, has_mul;    // GEMM: bias / mulY are real (absent ones point at the weights: fc_fetch loads unconditionally)
    int ldmy, mul_act;
    float mul_slope;
    // LOAD: source;  REPARAM: eps;  REPARAM_BWD also mu, logvar, g_mu, g_lv, gscale
    const float* src;
    int ld_src;
    const float* mu;
    const float* lv;
    const float* g_mu;
    const float* g_lv;
    const float* gscale;
    int kl_grad;
};

// work items of a wave, in order: (kind << 12) | (stage << 8) | tile
enum { FC_IT_TILE = 0, FC_IT_ELEM = 1, FC_IT_BARRIER = 2 };
struct FcChain {
    FcStage st[FC_MAX_STAGES];
    uint32_t items[FC_WAVES][FC_MAX_ITEMS / 2];  // two 16-bit items per dword (dwords are scalar loads; bytes and
    uint32_t n_items[FC_WAVES];                  // shorts would be vector loads with a vmcnt(0) wait each)
    int n, B;
    int lds_floats;
};
static_assert(sizeof(FcChain) <= 4096, "FcChain is passed by value as the kernel argument");

// Stage fields are read from the kernel-argument segment with a run-time stage index.  Left alone the compiler sinks
// every such scalar load into the (lane-masked) block that uses it: one s_load + s_waitcnt round trip in front of
// every vector load.  Passing the value through readfirstlane (a convergent operation: it cannot be sunk into
// divergent control flow) pins the loads to the top of the chunk, where they are issued together.
__device__ __forceinline__ int fc_u(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float fc_u(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// (pointers: rebuilt from their two uniform halves as GLOBAL pointers; a generic pointer made from an integer would
// turn every access into a flat_load, which also counts as an LDS operation in s_waitcnt)
typedef const __attribute__((address_space(1))) float* fc_gcp;
typedef __attribute__((address_space(1))) float* fc_gp;
__device__ __forceinline__ uint64_t fc_u64(const void* p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ fc_gcp fc_u(const float* p) { return (fc_gcp)fc_u64(p); }
__device__ __forceinline__ fc_gp fc_u(float* p) { return (fc_gp)fc_u64(p); }

// Loads and stores of fc_chain_k address memory as (wave-uniform base pointer) + (32-bit byte offset of the lane): the
// instruction takes the base from SGPRs and ONE VGPR, and the offset costs one or two 32-bit VALU operations instead
// of a 64-bit multiply-add per access.  (Measured: four waves share a SIMD, and the address arithmetic of the weight
// loads was the largest part of the kernel's VALU instructions.)  fc_finish checks the ranges: every byte offset below
// 2^31, rows and leading dimensions (in bytes) below 2^24 for the 24-bit multiplies.
typedef const __attribute__((address_space(1))) char* fc_gcb;
typedef __attribute__((address_space(1))) char* fc_gb;
__device__ __forceinline__ float fc_ld(fc_gcp base, uint32_t boff) {
    return *reinterpret_cast<fc_gcp>(reinterpret_cast<fc_gcb>(base) + boff);
}
__device__ __forceinline__ f32x4 fc_ld4(fc_gcp base, uint32_t boff) {
    return *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(reinterpret_cast<fc_gcb>(base) + boff);
}
__device__ __forceinline__ void fc_st(fc_gp base, uint32_t boff, float v) {
    *reinterpret_cast<fc_gp>(reinterpret_cast<fc_gb>(base) + boff) = v;
}

// The weights of one chunk (k-steps s0 .. s0 + FC_U of segment seg of the tile `item`) and the tile's bias and act'(Y)
// inputs.  Every load is unconditional, with clamped offsets: the s_waitcnt in front of the MFMAs can then count the
// loads of the NEXT chunk that may stay in flight (a load under a branch makes that count unknown and the wait a
// vmcnt(0), which serialises the prefetch -- which is also why an absent bias / act'(Y) is a load of offset 0 through
// a MASK: the select it used to be was compiled into a branch whose short side waited for vmcnt(0)), and nothing has
// to be zeroed: a clamped k meets zero padding on the A side, a clamped column is not stored.  VEC: some GEMM stage of
// the run has the reduction index contiguous in memory (wsk == 1, K % 4 == 0, aligned rows: st.vec) and takes one
// dwordx4 per k-step; else four dwords per k-step.
template <bool VEC>
__device__ __forceinline__ void fc_fetch(const FcChain& ch, int item, int seg, int s0, float4 (&b)[FC_U], float& bias,
                                         float (&my)[4], int m0, int li, int lk) {
    const FcStage& st = ch.st[(item >> 8) & 15];
    const int K = fc_u(st.K), N = fc_u(st.N);
    const uint32_t wsk4 = 4u * (uint32_t)fc_u((int)st.wsk), wsn4 = 4u * (uint32_t)fc_u((int)st.wsn);
    const fc_gcp W = fc_u(seg ? st.W2 : st.W);
    const fc_gcp bp = fc_u(st.bias);  // (never null: the host points absent ones at the weights, has_bias / has_mul = 0)
    const fc_gcp yp = fc_u(st.mulY);
    const uint32_t ldmy4 = 4u * (uint32_t)fc_u(st.ldmy);
    const int B = fc_u(ch.B);
    const uint32_t bias_mask = (uint32_t)-fc_u(st.has_bias), mul_mask = (uint32_t)-fc_u(st.has_mul);
    const uint32_t n = (uint32_t)min((item & 255) * 16 + li, N - 1);
    const uint32_t woff = __umul24(n, wsn4);
    // (the five small loads first: behind the weight loads, the compiler protected their address temporaries with a
    // wait that also drained the weights just requested)
    bias = fc_ld(bp, (4u * n) & bias_mask);
    const uint32_t n4m = (4u * n) & mul_mask, ldm = ldmy4 & mul_mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t m = (uint32_t)min(m0 + 4 * lk + r, B - 1);
        my[r] = fc_ld(yp, __umul24(m, ldm) + n4m);
    }
#if defined(FC_DIAG) && FC_DIAG == 1  // diagnostic build: no weight traffic
    for (int u = 0; u < FC_U; ++u) b[u] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (false)
#endif
    if (VEC && fc_u(st.vec)) {  // (VEC kernel, stage without it: the dword path below; the waits then count the
                                // loads of this path, which holds that stage's prefetch back, not the others')
        const uint32_t k0 = woff + 16u * (uint32_t)lk + 64u * (uint32_t)s0, kmax = woff + 4u * (uint32_t)(K - 4);
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const f32x4 t = fc_ld4(W, min(k0 + 64u * u, kmax));
            b[u] = make_float4(t[0], t[1], t[2], t[3]);
        }
    } else {
        const uint32_t k0 = woff + __umul24(4u * (uint32_t)lk, wsk4), kmax = woff + (uint32_t)(K - 1) * wsk4;
        const uint32_t sk = 16u * (uint32_t)s0 * wsk4;  // (uniform: the scalar unit's arithmetic)
#pragma unroll
        for (int u = 0; u < FC_U; ++u) {
            const uint32_t su = sk + 16u * u * wsk4;
            b[u].x = fc_ld(W, min(k0 + su, kmax));
            b[u].y = fc_ld(W, min(k0 + (su + wsk4), kmax));
            b[u].z = fc_ld(W, min(k0 + (su + 2u * wsk4), kmax));
            b[u].w = fc_ld(W, min(k0 + (su + 3u * wsk4), kmax));
        }
    }
}

// element-wise stages: every thread of the workgroup takes part
__device__ __forceinline__ void fc_elem(const FcChain& ch, const FcStage& st, float* fc_lds, int tid, int m0) {
    const int kind = fc_u(st.kind), N = fc_u(st.N), B = fc_u(ch.B);
    const int ld_src = fc_u(st.ld_src), ldo = fc_u(st.ldo);
    const int in_off = fc_u(st.in_off), in_stride = fc_u(st.in_stride), in2_off = fc_u(st.in2_off), in2_stride = fc_u(st.in2_stride);
    const int out_off = fc_u(st.out_off), out_stride = fc_u(st.out_stride), out2_off = fc_u(st.out2_off), out2_stride = fc_u(st.out2_stride);
    const fc_gcp src = fc_u(st.src);
    if (kind == FC_LOAD) {
        const fc_gcp yp = fc_u(st.mulY);
        const int ldmy = fc_u(st.ldmy), mul_act = fc_u(st.mul_act);
        const float mul_slope = fc_u(st.mul_slope);
        for (int e = tid; e < FC_ROWS * N; e += FC_THREADS) {
            const int row = e / N, c = e - row * N, m = m0 + row;
            float v = 0.f;
            if (m < B) {
                v = src[(int64_t)m * ld_src + c];
                if (yp != nullptr) v *= act_grad_from_out(yp[(int64_t)m * ldmy + c], mul_act, mul_slope);
            }
            fc_lds[out_off + row * out_stride + c] = v;
        }
    } else if (kind == FC_REPARAM_FWD) {
        const fc_gp out = fc_u(st.out);
        for (int e = tid; e < FC_ROWS * N; e += FC_THREADS) {
            const int row = e / N, c = e - row * N, m = m0 + row;
            if (m < B) {
                const float mu = fc_lds[in_off + row * in_stride + c];
                const float lv = fc_lds[in2_off + row * in2_stride + c];
                const float z = fmaf(src[(int64_t)m * ld_src + c], expf(0.5f * lv), mu);
                out[(int64_t)m * ldo + c] = z;
                if (out_off >= 0) fc_lds[out_off + row * out_stride + c] = z;
            }
        }
    } else {  // FC_REPARAM_BWD (elem.h: reparam_bwd_k)
        const fc_gp out = fc_u(st.out), out2 = fc_u(st.out2);
        const fc_gcp mup = fc_u(st.mu), lvp = fc_u(st.lv), g_mu = fc_u(st.g_mu), g_lv = fc_u(st.g_lv);
        const fc_gcp gscale = fc_u(st.gscale);
        const int kl_grad = fc_u(st.kl_grad);
        const float gs = kl_grad ? (gscale != nullptr ? gscale[0] : 1.f) : 0.f;
        for (int e = tid; e < FC_ROWS * N; e += FC_THREADS) {
            const int row = e / N, c = e - row * N, m = m0 + row;
            if (m < B) {
                const int64_t g = (int64_t)m * ld_src + c;
                const float lv = lvp[g], d = fc_lds[in_off + row * in_stride + c];
                const float sd = expf(0.5f * lv);
                float a = d, b = d * src[g] * 0.5f * sd;
                if (g_mu != nullptr) {
                    a += g_mu[g];
                    b += g_lv[g];
                }
                if (kl_grad) {
                    a = fmaf(gs, mup[g], a);
                    b = fmaf(gs, 0.5f * (sd * sd - 1.f), b);
                }
                out[(int64_t)m * ldo + c] = a;
                out2[(int64_t)m * ldo + c] = b;
                if (out_off >= 0) fc_lds[out_off + row * out_stride + c] = a;
                if (out2_off >= 0) fc_lds[out2_off + row * out2_stride + c] = b;
            }
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(FC_THREADS) void fc_chain_k(FcChain ch) {
    extern __shared__ __attribute__((aligned(16))) float fc_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * FC_ROWS;
    const int nit = fc_u((int)ch.n_items[wid]);
    const int B = fc_u(ch.B);
    IG_TDECL;
    IG_T();
    auto item_at = [&](int i) { return (int)((fc_u((int)ch.items[wid][i >> 1]) >> ((i & 1) * 16)) & 0xffff); };
    auto next_tile = [&](int i) {  // the first TILE item behind item i (nit: none)
        for (++i; i < nit && (item_at(i) >> 12) != FC_IT_TILE; ++i) {}
        return i;
    };
    // the chunk in flight: item ci, segment cseg, k-steps from cs0
    int ci = next_tile(-1), cseg = 0, cs0 = 0;
    float4 P[FC_U], Q[FC_U];
    float pb = 0.f, pm[4] = {0.f, 0.f, 0.f, 0.f}, qb = 0.f, qm[4] = {0.f, 0.f, 0.f, 0.f};
    int done = 0;  // items [0, done) are finished
    // A LOAD stage at the head of the list goes through registers: its loads are issued first, the first weights
    // behind them (a wave's loads return in order: the other way round the rows would wait for the weights).
    constexpr int LR = 5;
    float lr[LR], ly[LR];
    const int it0 = nit > 0 ? item_at(0) : (FC_IT_BARRIER << 12);
    const FcStage& ld0 = ch.st[(it0 >> 8) & 15];
    const int l_N = fc_u(ld0.N), l_ld = fc_u(ld0.ld_src), l_ldmy = fc_u(ld0.ldmy);
    const fc_gcp l_src = fc_u(ld0.src), l_y = fc_u(ld0.mulY);
    const bool reg_load = (it0 >> 12) == FC_IT_ELEM && fc_u(ld0.kind) == FC_LOAD && FC_ROWS * l_N <= LR * FC_THREADS;
    if (reg_load) {  // (unconditional loads with clamped indices, as in fc_fetch: the wait for them can then leave the
                     // weights requested behind them in flight)
        const fc_gcp yq = l_y != nullptr ? l_y : l_src;
        const int ldy = l_y != nullptr ? l_ldmy : l_ld;
#pragma unroll
        for (int q = 0; q < LR; ++q) {
            const int e = min(tid + q * FC_THREADS, FC_ROWS * l_N - 1);
            const int row = e / l_N, c = e - row * l_N, m = min(m0 + row, B - 1);
            lr[q] = l_src[(int64_t)m * l_ld + c];
            ly[q] = yq[(int64_t)m * ldy + c];
        }
    }
    // the padding columns stay zero (16 bytes per lane and store: every buffer is 16 rows of a multiple of 4 floats)
    for (int e = 4 * tid; e < fc_u(ch.lds_floats); e += 4 * FC_THREADS)
        *reinterpret_cast<float4*>(fc_lds + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ci < nit) fc_fetch<VEC>(ch, item_at(ci), 0, 0, P, pb, pm, m0, li, lk);
    lds_barrier();
    IG_T();
    if (reg_load) {
        const int o_off = fc_u(ld0.out_off), o_str = fc_u(ld0.out_stride), m_act = fc_u(ld0.mul_act);
        const float m_slope = fc_u(ld0.mul_slope);
#pragma unroll
        for (int q = 0; q < LR; ++q) {
            const int e = tid + q * FC_THREADS;
            const int row = e / l_N, c = e - row * l_N;
            if (e < FC_ROWS * l_N) {
                float v = m0 + row < B ? lr[q] : 0.f;
                if (l_y != nullptr) v *= act_grad_from_out(ly[q], m_act, m_slope);
                fc_lds[o_off + row * o_str + c] = v;
            }
        }
        done = 1;
    }
    auto run_markers = [&](int upto) {
        for (; done < upto; ++done) {
            const int it = item_at(done);
            if ((it >> 12) == FC_IT_BARRIER) {
                lds_barrier();
                IG_T();
            } else if ((it >> 12) == FC_IT_ELEM) {
                fc_elem(ch, ch.st[(it >> 8) & 15], fc_lds, tid, m0);
            }
        }
    };
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // One chunk: C holds its weights (fetched one step ago); the next chunk's go into Nx before the MFMAs start.  The
    // two register sets swap roles from step to step (two steps per loop iteration): no copy, so nothing waits for
    // the newer set before its own step.
    auto step = [&](const float4 (&C)[FC_U], const float& cb, const float (&cm)[4], float4 (&Nx)[FC_U], float& nb,
                    float (&nm)[4]) {
        const int item = item_at(ci);
        const FcStage& st = ch.st[(item >> 8) & 15];
        const int K = fc_u(st.K), in2_off = fc_u(st.in2_off);
        const int a_off = fc_u(cseg ? in2_off : st.in_off), a_stride = fc_u(cseg ? st.in2_stride : st.in_stride);
        const int nk = (K + 15) >> 4;
        // the chunk behind this one: of this tile, or the first of the wave's next tile
        int ni = ci, nseg = cseg, ns0 = cs0 + FC_U;
        if (ns0 >= nk) {
            ns0 = 0;
            if (++nseg >= (in2_off >= 0 ? 2 : 1)) {
                nseg = 0;
                ni = next_tile(ci);
            }
        }
        if (cseg == 0 && cs0 == 0) {  // first chunk of a tile: whatever the list holds before it (barriers, element-wise)
            run_markers(ci);
            acc0 = acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // (nothing with a load of its own between this fetch and the MFMAs, or their wait becomes a vmcnt(0))
        const bool more = ni < nit;  // (behind the last chunk: fetched again, unused, so that the load count is fixed)
        fc_fetch<VEC>(ch, item_at(more ? ni : ci), more ? nseg : cseg, more ? ns0 : cs0, Nx, nb, nm, m0, li, lk);
        const float* ap = fc_lds + a_off + li * a_stride + 4 * lk;
        // three k-steps of A per round of LDS reads (rows are zero-padded to a multiple of 48 columns, so the reads of a
        // short last group stay inside the row; its MFMAs are skipped)
#pragma unroll
        for (int g = 0; g < FC_U; g += 3) {
#if defined(FC_DIAG) && FC_DIAG == 2  // diagnostic build: no MFMA, no LDS reads
            for (int u = g; u < g + 3; ++u)
                if (cs0 + u < nk) { acc0[0] += C[u].x + C[u].y; acc1[0] += C[u].z + C[u].w; }
            if (false)
#endif
            if (cs0 + g < nk) {
                const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * (cs0 + g));
                const float4 a1 = *reinterpret_cast<const float4*>(ap + 16 * (cs0 + g + 1));
                const float4 a2 = *reinterpret_cast<const float4*>(ap + 16 * (cs0 + g + 2));
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, C[g].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, C[g].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, C[g].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, C[g].w, acc1, 0, 0, 0);
                if (cs0 + g + 1 < nk) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, C[g + 1].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, C[g + 1].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, C[g + 1].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, C[g + 1].w, acc1, 0, 0, 0);
                }
                if (cs0 + g + 2 < nk) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, C[g + 2].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, C[g + 2].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, C[g + 2].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, C[g + 2].w, acc1, 0, 0, 0);
                }
            }
        }
        if (ni != ci) {  // the tile is complete.  D: column li, rows 4 lk + r
            const int N = fc_u(st.N), act = fc_u(st.act), accumulate = fc_u(st.accumulate);
            const uint32_t ldo4 = 4u * (uint32_t)fc_u(st.ldo);
            const int out_off = fc_u(st.out_off), out_stride = fc_u(st.out_stride), mul_act = fc_u(st.mul_act);
            const float slope = fc_u(st.slope), mul_slope = fc_u(st.mul_slope);
            const fc_gp out = fc_u(st.out);
            const bool has_mul = fc_u(st.has_mul) != 0, has_bias = fc_u(st.has_bias) != 0;
            const int n = (item & 255) * 16 + li;
            const bool n_ok = n < N;
            const float bv = has_bias ? cb : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * lk + r, m = m0 + row;
                const bool ok = n_ok && m < B;
                float v = act_apply(acc0[r] + acc1[r] + bv, act, slope);
                if (ok) {
                    const uint32_t go = __umul24((uint32_t)m, ldo4) + 4u * (uint32_t)n;
                    if (accumulate) v += fc_ld(out, go);
                    fc_st(out, go, v);
                }
                if (out_off >= 0) {
                    float l = ok ? v : 0.f;
                    if (has_mul) l *= act_grad_from_out(cm[r], mul_act, mul_slope);
                    fc_lds[out_off + row * out_stride + n] = l;
                }
            }
            done = ci + 1;
        }
        ci = ni; cseg = nseg; cs0 = ns0;
    };
    while (ci < nit) {
        step(P, pb, pm, Q, qb, qm);
        if (ci >= nit) break;
        step(Q, qb, qm, P, pb, pm);
    }
    run_markers(nit);
#ifdef IG_TIMERS
    if (threadIdx.x == 0 && blockIdx.x == 1) printf("fc_chain %d stages: zero | levels ...", ch.n);
#endif
    IG_TPRINT("");
}
