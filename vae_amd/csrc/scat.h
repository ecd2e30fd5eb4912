// scat.h - transposed 3x3 stride-2 gathers of the mid-size layers as a dense GEMM followed by col2im ("scatter" form):
// ConvTranspose2d forward (convT1 32->16 3x3 -> 7x7, convT2 16->8 7x7 -> 14x14 of the notebook model) and the input
// gradient of a stride-2 Conv2d (conv3, conv2), which is the same operation with the channel roles swapped.
//
// As a gather (igemm_gather_k) every output-parity class is its own small GEMM with 1, 2, 2 or 4 taps, 49-pixel classes
// fill 3 1/16 of 4 pixel tiles, 8 output channels fill half an MFMA's rows, and the B operand is gathered through a
// tap table: 144 MFMAs per sample for convT2 and a dependent table -> operand chain.  Here the product is taken over
// the INPUT pixels instead:   Y[(tap, co)][pixel] = sum_ci W[ci][co][tap] * x[ci][pixel]
// - a plain GEMM (M = 9*Cout rows, N = the round's input pixels, K = Cin) whose B operand is the input image as it lies
// in memory (no gather, no halo), 80 MFMAs per sample for convT2 - and every output pixel then adds the <= 4 products
// that land on it (oy = 2*iy - ph + kh) from an LDS tile [pixel][(tap, co)], eight or sixteen channels per 16-byte
// read.  Streaming like thin.h / mid_conv_k: a 256-thread workgroup walks rounds of SB samples (<= 64 input pixels),
// the loads of the round two ahead are in flight, weights (A operand) stay in registers for the whole kernel.
//   IN = 0: producer BatchNorm (+ReLU) or nothing on load; IN = 1: the input is a gradient,
//           (ga*g + gb*(y - mean) + gc) * act'(y) on load (ACT = that layer's activation)
//   EP = 0: bias + activation (ACT) and, with a slab, forward statistics {pivot = bias, sum(y-p), sum((y-p)^2), n};
//   EP = 1: nothing; EP = 2: backward through the producer's virtual BatchNorm (+ReLU): mask, and its two sums per
//           channel.  One slab row per workgroup, and the BatchNorm's fixed-point accumulators (igemm.h).
// Uses IgArgs (igemm.h) like mid_conv_k: geometry (of the transposed gather), tensors, epilogue; SB = samples per round.
#pragma once
#include <type_traits>

#include "thin.h"

#define SC_BP 80    // row stride of the B image: 64 pixel columns + 16, so that the four k rows of an MFMA step hit different banks
#define SC_ZROW 64  // the all-zero row of the product tile that absent taps read
// row stride of the product tile [pixel][(tap, co)]: the row tiles + 4 floats, = 20 (mod 64): the sixteen pixel lanes of
// a 16-byte store land in sixteen different bank quads (with a multiple of 16 floats they shared four)
#define SC_MP(CO) (16 * ((9 * (CO) + 15) / 16) + 4)

template <int CO, int KS, int NJ, int IN, int EP, int ACT>
__global__ __launch_bounds__(TH_THREADS, 2) void scat_convT_k(IgArgs a) {  // (two workgroups per CU: <= 256 registers, accumulation registers included)
    constexpr int MT = (9 * CO + 15) / 16, MP = SC_MP(CO), KR = 4 * KS, CV = CO / 4;
    // who owns which product tiles: few row tiles (MT <= 5): wave w takes the pixel tile w and every row tile; many:
    // wave w takes the row tiles w, w + 4, ... of every pixel tile (a third of the weight registers)
    constexpr bool MSPLIT = MT > 5;
    constexpr int MTW = MSPLIT ? (MT + 3) / 4 : MT;
    // PIPE: two product tiles; between two barriers a wave takes the products of round r + 1 (matrix pipe) and the
    // col2im of round r (LDS reads, vector ALU, stores): one barrier per round instead of two, and the two kinds of
    // work overlap.  (Not for the 16-channel instances: their tiles are 38 KB each and a workgroup has 2 rounds.)
    constexpr bool PIPE = !MSPLIT;
    constexpr int YF = (SC_ZROW + 1) * MP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const IgGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int HW = g.H * g.W, CHW = g.Cin * HW, OHW = g.OH * g.OW, SBR = a.SB, N = g.N;
    float* bm[2] = {smem, smem + KR * SC_BP};   // B image [k = ci][pixel of the round], double-buffered
    float* yf = smem + 2 * KR * SC_BP;          // products [pixel][(tap, co)] + the zero row (PIPE: two of them)
    float* cin = yf + (PIPE ? 2 : 1) * YF;      // input coefficients [4 * Cin + 4]
    float* cep = cin + 4 * g.Cin + 4;           // EP 2: producer BatchNorm [CO][4] + threshold; EP 0: bias at [4 * CO + 4, 5 * CO + 4)
    float* red = cep + 5 * CO + 4;              // [4][2 * CO] + [2 * CO]
    const int G = gridDim.x, last = a.B - 1;
    const int nrounds = (a.B + SBR - 1) / SBR;
    const int vps = CHW >> 2, nvec = SBR * vps;  // float4 per sample / per round (the host checks 4 | CHW)

    IG_TDECL;
    IG_T();
    BnFinRegs finr;
    if (IN == 0) bnfin_issue<false>(a.fin, finr);

    // ---- staging slots: float4 e4 = tid + 256 j of the round's [sample][ci][pixel] block
    int s_l[NJ], rem[NJ];
    bool v_ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e4 = tid + 256 * j;
        v_ok[j] = e4 < nvec;
        const int e4c = v_ok[j] ? e4 : nvec - 1;
        s_l[j] = e4c / vps;
        rem[j] = (e4c - s_l[j] * vps) * 4;
    }
    f32x4 vA[NJ], vB[NJ], yA[IN ? NJ : 1], yB[IN ? NJ : 1];
    auto load = [&](int r, f32x4 (&v)[NJ], f32x4 (&y)[IN ? NJ : 1]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int s = min(r * SBR + s_l[j], last);
            const int64_t o = (int64_t)s * CHW + rem[j];
            v[j] = *reinterpret_cast<const f32x4*>(a.in.src + o);
            if (IN) y[j] = *reinterpret_cast<const f32x4*>(a.in.src2 + o);
        }
    };
    load(blockIdx.x, vA, yA);
    load(blockIdx.x + G, vB, yB);
    IG_T();

    // ---- once per workgroup
    // What has to be zero: the k rows >= Cin of both B images (their weights are zero, but 0 * garbage may be NaN) and
    // the zero row of the product tile.  Pixel columns past the round only feed product columns that nobody reads.
    for (int e = g.Cin * SC_BP + tid; e < KR * SC_BP; e += TH_THREADS) bm[0][e] = bm[1][e] = 0.f;
    for (int e = tid; e < MP; e += TH_THREADS) yf[SC_ZROW * MP + e] = yf[(PIPE ? YF : 0) + SC_ZROW * MP + e] = 0.f;
    if (IN) {
        for (int e = tid; e < 4 * g.Cin; e += TH_THREADS) {
            const int q = e / g.Cin, c = e - q * g.Cin;
            float v = q == 1 ? 1.f : 0.f;
            if (a.in.mode == TR_GRAD_BN) v = a.in.coef[(q == 0 ? 0 : 3 + q) * g.Cin + c];
            cin[e] = v;
        }
    } else if (a.fin.C == 0) {
        for (int e = tid; e < 4 * g.Cin; e += TH_THREADS) {
            float v = (e >= 2 * g.Cin && e < 3 * g.Cin) ? 1.f : 0.f;
            if (a.in.mode == TR_BN) v = a.in.coef[e];
            cin[e] = v;
        }
    }
    if (!IN && tid == 0) cin[4 * g.Cin] = (a.in.mode == TR_BN && a.in.relu) ? 0.f : -__builtin_inff();
    if (EP == 2) {  // one 16-byte row {mean, rstd, scale, beta} per produced channel (rows past N repeat the last one)
        for (int e = tid; e < 4 * CO; e += TH_THREADS) cep[e] = a.ep_coef[(e & 3) * N + min(e >> 2, N - 1)];
        if (tid == 0) cep[4 * CO] = a.ep_relu ? 0.f : -__builtin_inff();
    }
    if (EP == 0 && tid < CO) cep[4 * CO + 4 + tid] = (a.bias != nullptr && tid < N) ? a.bias[tid] : 0.f;
    IG_T();
    int dst[NJ][4], chn[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = rem[j] + u;
            const int ci = (int)fdiv((uint32_t)e, g.d_HW), p = e - ci * HW;
            chn[j][u] = ci;
            dst[j][u] = ci * SC_BP + s_l[j] * HW + p;
        }
    // weights: A operand, row m = 16 mt + li = (tap, co), k = 4 s + lk = ci; every load is issued before the first mask
    float wreg[MTW][KS];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int mt = MSPLIT ? wid + 4 * i : i;
        const int m = 16 * mt + li, tap = m / CO, co = m - tap * CO;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wreg[i][s] = a.w[min(4 * s + lk, g.Cin - 1) * g.w_cstride + min(co, N - 1) * g.w_nstride + min(tap, 8)];
    }
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int mt = MSPLIT ? wid + 4 * i : i;
        const int m = 16 * mt + li, tap = m / CO, co = m - tap * CO;
#pragma unroll
        for (int s = 0; s < KS; ++s) wreg[i][s] = (tap < 9 && co < N && 4 * s + lk < g.Cin) ? wreg[i][s] : 0.f;
    }
    IG_T();
    // col2im: this thread's (sample of the round, output pixel) and the <= 4 products that land on it
    const bool q_ok = tid < SBR * OHW;
    const int q_c = q_ok ? tid : 0;
    const int q_s = q_c / OHW, q_p = q_c - q_s * OHW;
    int ent[4];
    {
        const int oy = q_p / g.OW, ox = q_p - oy * g.OW;
        int iy0 = -1, iy1 = -1, kh0 = 0, kh1 = 0, ix0 = -1, ix1 = -1, kw0 = 0, kw1 = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int th = oy + g.ph - k, tw = ox + g.pw - k;
            if (th >= 0 && !(th & 1) && (th >> 1) < g.H) {
                if (iy0 < 0) { iy0 = th >> 1; kh0 = k; } else { iy1 = th >> 1; kh1 = k; }
            }
            if (tw >= 0 && !(tw & 1) && (tw >> 1) < g.W) {
                if (ix0 < 0) { ix0 = tw >> 1; kw0 = k; } else { ix1 = tw >> 1; kw1 = k; }
            }
        }
        const int zero = SC_ZROW * MP;
        auto entry = [&](int iy, int kh, int ix, int kw) {
            return (iy >= 0 && ix >= 0) ? (q_s * HW + iy * g.W + ix) * MP + (kh * 3 + kw) * CO : zero;
        };
        ent[0] = entry(iy0, kh0, ix0, kw0); ent[1] = entry(iy0, kh0, ix1, kw1);
        ent[2] = entry(iy1, kh1, ix0, kw0); ent[3] = entry(iy1, kh1, ix1, kw1);
    }
    float s1[CO], s2[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) s1[c] = s2[c] = 0.f;
    IG_T();
    __syncthreads();
    if (IN == 0 && a.fin.C > 0) {
        // every workgroup finalises the producer BatchNorm from its accumulators (requested at the top); the scratch
        // lies over the product rows, which are written before they are read
        static_assert(SC_ZROW * MP >= BNFIN_SCR_FLOATS, "the finalize prologue's scratch lies over the product tile");
        bnfin_forward<TH_THREADS>(a.fin, finr, cin, yf);  // (ends with a barrier)
    }
    float bias_r[(EP == 0 && CO <= 8) ? CO : 1];  // EP 0, 8 channels: the bias of every produced channel (zero past N) in registers
    if (EP == 0 && CO <= 8) {
#pragma unroll
        for (int c = 0; c < CO; ++c) bias_r[c] = cep[4 * CO + 4 + c];
    }
    const float thr_in = IN ? 0.f : cin[4 * g.Cin];
    const float thr_ep = EP == 2 ? cep[4 * CO] : 0.f;

    auto store = [&](float* t, const f32x4 (&v)[NJ], const f32x4 (&y)[IN ? NJ : 1]) __attribute__((always_inline)) {
        float o[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = chn[j][u];
                if (IN)
                    o[j][u] = fmaf(cin[g.Cin + c], v[j][u], fmaf(cin[2 * g.Cin + c], y[j][u] - cin[c], cin[3 * g.Cin + c])) *
                              th_act_grad<ACT>(y[j][u], a.in.slope);
                else
                    o[j][u] = fmaxf(fmaf(v[j][u] - cin[c], cin[2 * g.Cin + c], cin[3 * g.Cin + c]), thr_in);
            }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (v_ok[j]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) t[dst[j][u]] = o[j][u];
            }
    };
    const int ntl = (SBR * HW + 15) >> 4;  // pixel tiles that hold pixels of the round
    auto products = [&](const float* t, float* yo) __attribute__((always_inline)) {
        if (MSPLIT) {
            for (int nt = 0; nt < ntl; ++nt) {
                const float* p = t + lk * SC_BP + 16 * nt + li;
                float b[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) b[s] = p[4 * s * SC_BP];
                float* o = yo + (16 * nt + li) * MP + 4 * lk;
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    if (wid + 4 * i >= MT) continue;  // (wave-uniform)
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i][s], b[s], acc, 0, 0, 0);
                    *reinterpret_cast<f32x4*>(o + 16 * (wid + 4 * i)) = acc;
                }
            }
        } else {
            if (wid >= ntl) return;
            const float* p = t + lk * SC_BP + 16 * wid + li;
            float b[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) b[s] = p[4 * s * SC_BP];
            float* o = yo + (16 * wid + li) * MP + 4 * lk;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[mt][s], b[s], acc, 0, 0, 0);
                *reinterpret_cast<f32x4*>(o + 16 * mt) = acc;  // D: rows lk*4 + r = (tap, co), column li = pixel
            }
        }
    };
    float ev[EP == 2 ? CO : 1];
    auto ep_load = [&](int r) __attribute__((always_inline)) {  // EP 2: the producer's outputs at this thread's output pixel, requested before the products
        if (EP != 2) return;
        const int s = min(r * SBR + q_s, last);
        // (opaque pixel index: otherwise every channel's 64-bit address is hoisted out of the round loop, once per
        // inlined copy, and the 16-channel instances run out of registers)
        int qp = q_p;
        asm volatile("" : "+v"(qp));
        const float* py = a.ep_y + (int64_t)s * N * OHW + qp;
#pragma unroll
        for (int c = 0; c < CO; ++c) ev[c] = py[min(c, N - 1) * OHW];
    };
    const bool nfull = CO <= 8 && N == CO;  // (uniform) every produced channel exists: no masks on the stores (8-channel instances: the 16-channel ones have no registers for a second copy of the epilogue)
    // (always_inline: called twice, the compiler left it out of line and its captured arrays went to scratch memory)
    auto scatter_f = [&](int r, const float* yi, auto full) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full)::value;
        const int s = r * SBR + q_s;
        if (!q_ok || s > last) return;
        f32x4 sum[CV];
#pragma unroll
        for (int h = 0; h < CV; ++h) sum[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4* y = reinterpret_cast<const f32x4*>(yi + ent[e]);
#pragma unroll
            for (int h = 0; h < CV; ++h) sum[h] += y[h];
        }
        int qp = q_p;
        asm volatile("" : "+v"(qp));  // (see ep_load)
        float* o = a.out + (int64_t)s * N * OHW + qp;
        // (no per-channel branches: a `continue` on c >= N put a branch and an LDS wait in front of every channel.
        // Channels >= N compute on zero products and clamped coefficients; only their stores are masked.)
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float acc = sum[c >> 2][c & 3];
            float v;
            if (EP == 2) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(cep + 4 * c);
                const float xc = ev[c] - q[0];
                v = fmaf(xc, q[2], q[3]) > thr_ep ? acc : 0.f;
                s1[c] += v;
                s2[c] = fmaf(v, xc * q[1], s2[c]);
            } else if (EP == 1) {
                v = acc;
            } else {
                const float bs = CO <= 8 ? bias_r[c] : cep[4 * CO + 4 + c];
                v = th_act<ACT>(acc + bs, a.slope);
                const float d = v - bs;  // statistics pivot: the channel's bias
                s1[c] += d;
                s2[c] = fmaf(d, d, s2[c]);
            }
            if (FULL || c < N) o[c * OHW] = v;
            // (16 channels: without a fence every four channels the scheduler hoists all 64 coefficient reads and the
            // kernel needs > 256 registers: one workgroup per CU, 22.7 -> 30.8 us)
            if (CO > 8 && (EP == 2 || (c & 3) == 3)) asm volatile("" ::: "memory");
        }
    };
    auto scatter = [&](int r, const float* yi) __attribute__((always_inline)) {
        if (nfull) scatter_f(r, yi, std::true_type{});
        else scatter_f(r, yi, std::false_type{});
    };

    IG_T();
    store(bm[0], vA, yA);
    load(blockIdx.x + 2 * G, vA, yA);
    if (PIPE) {
        lds_barrier();
        products(bm[0], yf);
        store(bm[1], vB, yB);
        load(blockIdx.x + 3 * G, vB, yB);
        lds_barrier();
        for (int r = blockIdx.x; r < nrounds; r += 2 * G) {
            // image 1 and product tile 0 are complete; image 0 and product tile 1 are free
            IG_T();
            ep_load(r);
            products(bm[1], yf + YF);
            IG_T();
            scatter(r, yf);
            IG_T();
            store(bm[0], vA, yA);
            load(r + 4 * G, vA, yA);
            IG_T();
            lds_barrier();
            IG_T();
            ep_load(r + G);
            products(bm[0], yf);
            scatter(r + G, yf + YF);
            store(bm[1], vB, yB);
            load(r + 5 * G, vB, yB);
            lds_barrier();
        }
    } else {
        for (int r = blockIdx.x; r < nrounds; r += 2 * G) {
            lds_barrier();  // image 0 is complete, the product tile is free
            IG_T();
            ep_load(r);
            products(bm[0], yf);
            IG_T();
            store(bm[1], vB, yB);
            load(r + 3 * G, vB, yB);
            IG_T();
            lds_barrier();  // the products are complete
            IG_T();
            scatter(r, yf);
            IG_T();
            lds_barrier();
            ep_load(r + G);
            products(bm[1], yf);
            store(bm[0], vA, yA);
            load(r + 4 * G, vA, yA);
            lds_barrier();
            scatter(r + G, yf);
        }
    }
    IG_T();
    IG_TPRINT("scat: loads zero+coef dst+weights entries [sync/bnfin] | pipe: (prologue) products scatter store+load [barrier] (second round) ... | else: store0 [barrier] products store+load [barrier] scatter (second round) ...");

    if (a.slab == nullptr || EP == 1) return;  // (uniform)
    // ---- per-channel sums over the workgroup -> its slab row and the BatchNorm's accumulators
    float tot[2 * CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) { tot[c] = s1[c]; tot[CO + c] = s2[c]; }
    __syncthreads();
    th_block_sums<2 * CO>(tot, red, red + 8 * CO);
    if (tid < N) {
        const float t1 = red[8 * CO + tid], t2 = red[9 * CO + tid];
        float* sl = a.slab + ((int64_t)blockIdx.x * N + tid) * 4;
        if (EP == 2) {
            sl[0] = t1; sl[1] = t2; sl[2] = 0.f; sl[3] = 0.f;
            if (a.acc_save != nullptr) bnacc_add_bwd(a.acc_save, N, tid, t1, t2);
        } else {
            int ns = 0;  // samples this workgroup walked
            for (int r = blockIdx.x; r < nrounds; r += G) ns += min(SBR, a.B - r * SBR);
            const float bp = cep[4 * CO + 4 + tid], cn = (float)ns * (float)OHW;
            sl[0] = bp; sl[1] = t1; sl[2] = t2; sl[3] = cn;
            if (a.acc_save != nullptr) {
                const float dd = t1 / cn, pv = a.acc_pivot[tid];
                bnacc_add_fwd(a.acc_save, N, tid, cn, bp + dd, fmaxf(t2 - t1 * dd, 0.f), pv);
                if (blockIdx.x == 0) a.acc_save[7 * N + tid] = pv;
            }
        }
    }
}
