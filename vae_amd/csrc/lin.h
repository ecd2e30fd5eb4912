// lin.h - Linear layers on fp32 MFMA (v_mfma_f32_16x16x4_f32) for skinny shapes.
//
// The Linear layers on this path have M = batch (thousands) but K, N in 15..288, so a classic
// K-looped GEMM is latency-bound (a few k-steps of a few microseconds each on a handful of
// blocks).  Instead each workgroup takes a 32-row x 32-column output tile and loads its WHOLE K
// extent in one shot (every global load in flight at once: a single memory latency), then runs
// the MFMA chain out of LDS.
//
//   lin_fwd_k   : C[b][n] (=|+=) act( sum_k A'[b][k] * W'(k,n) + bias[n] )
//                 forward (A' = x, W'(k,n) = W[n][k]) and input gradient
//                 (A' = dY * act'(Y) taken on load, W'(k,n) = W[k][n]).
//   lin_wgrad_k : dW[n][k] += sum_b dY'[b][n] * X[b][k],  db[n] += sum_b dY'[b][n]
//                 (K of the MFMA = batch; 64 x 64 output tile per workgroup, a wave = 16 rows x 4 column tiles: the
//                 launch is bound by re-reads of X / dY through L2, which the wider tile cuts by a quarter: 34.5 ->
//                 31.4 us for the notebook model's five layers).  Every batch slice (workgroup column gx) writes its tile into its own
//                 row of the layer's partial region; wgrad_finalize_k adds the rows in a fixed order, so the
//                 gradient is bit-identical from run to run (no float atomics).
#pragma once
#include <type_traits>
#include "common.h"
#include "igemm.h"  // f32x4, FastDiv

#define LIN_THREADS 256
#define LIN_BM 32
#define LIN_BN 32
#define LIN_WS 33  // row stride of the [k][n] weight tile: conflict-free transposed fill

struct LinArgs {
    const float* A;   // [B][K], row stride lda
    const float* A2;  // optional [B][K] (row stride lda2): A' = A * act'(A2)
    const float* W;   // W'(k,n) at W[k*wsk + n*wsn]
    const float* bias;
    float* C;         // [B][N], row stride ldc
    int lda, lda2, ldc;
    int64_t wsk, wsn;
    int B, K, N;
    int KC;           // K extent staged per pass (multiple of 16); K is walked in ceil(K/KC) passes
    int Kp;           // KC + 1: odd row stride of the A tile
    int a_act, act, accumulate;
    float a_slope, slope;
    FastDiv d_KC, d_KC4;  // by KC and by KC/4
    // Two Linear layers that read the same input run as one launch:
    //  forward  - a second head (W2, bias2, C2, N2, ldc2) takes the column blocks blockIdx.y >= ny1 (0: no second head);
    //  backward - the input gradient sums both layers: a second K segment (Ab, A2b, Wb; same K, lda, lda2, strides)
    //             is walked after the first (nseg = 2).
    const float* W2;
    const float* bias2;
    float* C2;
    int N2, ldc2, ny1;
    const float* Ab;
    const float* A2b;
    const float* Wb;
    int nseg;
};

// VEC: rows are 16-byte aligned and K, N are multiples of 4 where it matters (host-checked), so the tile
// is fetched with dwordx4 loads: a quarter of the load instructions, all of a pass in flight at once.
template <bool VEC, bool BF = false>
__global__ __launch_bounds__(LIN_THREADS) void lin_fwd_k(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [32][Kp]
    float* Ws = As + LIN_BM * a.Kp;    // [KC][33]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    // (second forward head: its own weights, bias and output)
    const bool head2 = a.ny1 > 0 && (int)blockIdx.y >= a.ny1;
    const float* const Wh = head2 ? a.W2 : a.W;
    const float* const biash = head2 ? a.bias2 : a.bias;
    float* const Ch = head2 ? a.C2 : a.C;
    const int Nh = head2 ? a.N2 : a.N, ldch = head2 ? a.ldc2 : a.ldc;
    const int m0 = blockIdx.x * LIN_BM, n0 = (head2 ? blockIdx.y - a.ny1 : blockIdx.y) * LIN_BN;

    const int mt = wid >> 1, nt = wid & 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = As + (mt * 16 + li) * a.Kp + lk;
    const float* wp = Ws + lk * LIN_WS + nt * 16 + li;
    const int total = LIN_BM * a.KC;  // == LIN_BN * a.KC
    const bool nfast = (a.wsn == 1);
    for (int seg = 0; seg < a.nseg; ++seg) {
    const float* const Ag = seg ? a.Ab : a.A;
    const float* const A2g = seg ? a.A2b : a.A2;
    const float* const Wg = seg ? a.Wb : Wh;
    for (int kc0 = 0; kc0 < a.K; kc0 += a.KC) {
        if (kc0 > 0 || seg > 0) __syncthreads();
        if (VEC) {
            const int total4 = total >> 2;  // groups of 4 consecutive k (A, W[n][k]) or n (W[k][n])
            constexpr int U = 5;
            for (int base = tid; base < total4; base += LIN_THREADS * U) {
                float4 va[U], vy[U], vw[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int e = base + j * LIN_THREADS;
                    va[j] = vw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    vy[j] = va[j];
                    if (e < total4) {
                        const int r = fdiv((uint32_t)e, a.d_KC4), k = (e - r * (a.KC >> 2)) << 2;
                        if (m0 + r < a.B && kc0 + k < a.K) {
                            va[j] = *reinterpret_cast<const float4*>(Ag + (int64_t)(m0 + r) * a.lda + kc0 + k);
                            if (A2g != nullptr)
                                vy[j] = *reinterpret_cast<const float4*>(A2g + (int64_t)(m0 + r) * a.lda2 + kc0 + k);
                        }
                        const int wn = nfast ? (e & 7) << 2 : r, wk = nfast ? (e >> 3) : k;
                        if (n0 + wn < Nh && kc0 + wk < a.K)
                            vw[j] = *reinterpret_cast<const float4*>(Wg + (int64_t)(kc0 + wk) * a.wsk + (int64_t)(n0 + wn) * a.wsn);
                    }
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int e = base + j * LIN_THREADS;
                    if (e < total4) {
                        const int r = fdiv((uint32_t)e, a.d_KC4), k = (e - r * (a.KC >> 2)) << 2;
                        if (A2g != nullptr) {
                            va[j].x *= act_grad_from_out(vy[j].x, a.a_act, a.a_slope);
                            va[j].y *= act_grad_from_out(vy[j].y, a.a_act, a.a_slope);
                            va[j].z *= act_grad_from_out(vy[j].z, a.a_act, a.a_slope);
                            va[j].w *= act_grad_from_out(vy[j].w, a.a_act, a.a_slope);
                        }
                        float* ad = As + r * a.Kp + k;
                        ad[0] = va[j].x; ad[1] = va[j].y; ad[2] = va[j].z; ad[3] = va[j].w;
                        if (nfast) {
                            float* wd = Ws + (e >> 3) * LIN_WS + ((e & 7) << 2);
                            wd[0] = vw[j].x; wd[1] = vw[j].y; wd[2] = vw[j].z; wd[3] = vw[j].w;
                        } else {
                            float* wd = Ws + k * LIN_WS + r;
                            wd[0] = vw[j].x; wd[LIN_WS] = vw[j].y; wd[2 * LIN_WS] = vw[j].z; wd[3 * LIN_WS] = vw[j].w;
                        }
                    }
                }
            }
        } else
        for (int base = tid; base < total; base += LIN_THREADS * 8) {
            float va[8], vw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = base + j * LIN_THREADS;
                va[j] = 0.f;
                vw[j] = 0.f;
                if (e < total) {
                    const int r = fdiv((uint32_t)e, a.d_KC), k = e - r * a.KC;
                    if (m0 + r < a.B && kc0 + k < a.K) {
                        va[j] = Ag[(int64_t)(m0 + r) * a.lda + kc0 + k];
                        if (A2g != nullptr)
                            va[j] *= act_grad_from_out(A2g[(int64_t)(m0 + r) * a.lda2 + kc0 + k], a.a_act, a.a_slope);
                    }
                    // weight tile: walk the memory-contiguous index fastest
                    const int wn = nfast ? (e & 31) : r, wk = nfast ? (e >> 5) : k;
                    if (n0 + wn < Nh && kc0 + wk < a.K) vw[j] = Wg[(int64_t)(kc0 + wk) * a.wsk + (int64_t)(n0 + wn) * a.wsn];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = base + j * LIN_THREADS;
                if (e < total) {
                    const int r = fdiv((uint32_t)e, a.d_KC), k = e - r * a.KC;
                    As[r * a.Kp + k] = va[j];
                    const int wn = nfast ? (e & 31) : r, wk = nfast ? (e >> 5) : k;
                    Ws[wk * LIN_WS + wn] = vw[j];
                }
            }
        }
        __syncthreads();
        if constexpr (BF) {  // (igemm.h: bf16 compute mode) this lane's k-values of a 16-wide block: lk*4 + 0..3
            const float* apb = As + (mt * 16 + li) * a.Kp + lk * 4;
            const float* wpb = Ws + lk * 4 * LIN_WS + nt * 16 + li;
            for (int k4 = 0; k4 < a.KC; k4 += 16) {
                const s16x4 av = bf16_pack4(apb[k4], apb[k4 + 1], apb[k4 + 2], apb[k4 + 3]);
                const s16x4 bv = bf16_pack4(wpb[k4 * LIN_WS], wpb[(k4 + 1) * LIN_WS], wpb[(k4 + 2) * LIN_WS],
                                            wpb[(k4 + 3) * LIN_WS]);
                acc = MFMA_BF16(av, bv, acc);
            }
        } else
        for (int k4 = 0; k4 < a.KC; k4 += 16) {
            const float a0 = ap[k4], a1 = ap[k4 + 4], a2 = ap[k4 + 8], a3 = ap[k4 + 12];
            const float b0 = wp[k4 * LIN_WS], b1 = wp[(k4 + 4) * LIN_WS], b2 = wp[(k4 + 8) * LIN_WS],
                        b3 = wp[(k4 + 12) * LIN_WS];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
        }
    }
    }
    // D: col = li, row = lk*4 + r
    const int n = n0 + nt * 16 + li;
    if (n >= Nh) return;
    const float bv = biash != nullptr ? biash[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + mt * 16 + lk * 4 + r;
        if (m < a.B) {
            float v = act_apply(acc[r] + bv, a.act, a.slope);
            float* cp = Ch + (int64_t)m * ldch + n;
            *cp = a.accumulate ? *cp + v : v;
        }
    }
}

// ---- large batches: weights stay in LDS, rows stream ---------------------------------------------------------
// lin_fwd_k re-stages a 32-column slice of W and its 32 rows of A per workgroup: at B = 65,536 the 100 -> 288 layer is
// 18,432 one-shot workgroups that each wait out one load latency (0.8 TB/s, 115-150 us per launch; the Linear layers are
// then 27 % of the MNIST step).  lin_ws_k keeps a column group of W' (all of K, <= 304 columns: 115 KB for 288 x 100)
// in LDS for the whole kernel - one workgroup of 16 waves per CU - and walks 16-row tiles of A: the tile two... one ahead
// is in flight in registers while the current one is multiplied (wave w owns the column tiles w, w + 16 and
// shares every A operand read among them).  Same LinArgs and semantics as lin_fwd_k (second head, second K segment,
// act' on load, bias + activation, accumulate); fp32, 16-byte aligned rows only (the host checks).
// (16 waves: four per SIMD.  With 8 the k loop ran at half the matrix pipe's rate - 4-byte LDS operand reads reach their
// rate only at ~4 waves per SIMD, MI355X_MICROARCH.md, LDS - and staging / epilogue had nothing to overlap with)
#define LWS_THREADS 1024
#define LWS_WAVES 16
#define LWS_RT 16     // rows per tile
#define LWS_MAXNT 2   // column tiles per wave: <= 32 tiles = 512 columns per group
#define LWS_NJ 2      // float4 of a row tile per thread: 16 * Ktot / 4 <= 2048
struct LinWsPlan {
    int NC, S4, PW, PX, ng1, ng2;  // columns per group (multiple of 16); floats per (column | row) of one k-residue plane
                                   // (4 * odd >= k-steps rounded up to 4); plane sizes (multiples of 64 floats); column
                                   // groups of head 1 / head 2
};
// LDS: operands are laid out for 16-byte reads of FOUR consecutive k-steps: the lane (lk, li) of an MFMA step s reads
// k = 4 s + lk, so k-residue lk gets a plane of its own, Wl[lk][n][s] and Xl[lk][row][s] with s contiguous.  Row stride
// S4 = 4 * odd floats and planes that are multiples of 64 floats: the sixteen lanes of a ds_read_b128 group (all one
// residue) hit sixteen different 16-byte bank groups.  4-byte operand reads (one per lane and k-step) ran the k loop at
// half the matrix pipe's rate.  k-steps past the real ones are zero in both images.
__global__ __launch_bounds__(LWS_THREADS) void lin_ws_k(LinArgs a, LinWsPlan p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int Kt = a.K * a.nseg, K4 = a.K >> 2;
    float* Wl = smem;               // [4][NC][S4] (planes PW apart)
    float* Xl = smem + 4 * p.PW;    // [4][16][S4] (planes PX apart)
    const bool head2 = (int)blockIdx.y >= p.ng1;
    const float* const Wh = head2 ? a.W2 : a.W;
    const float* const biash = head2 ? a.bias2 : a.bias;
    float* const Ch = head2 ? a.C2 : a.C;
    const int Nh = head2 ? a.N2 : a.N, ldch = head2 ? a.ldc2 : a.ldc;
    const int n0 = (head2 ? (int)blockIdx.y - p.ng1 : (int)blockIdx.y) * p.NC;
    const int ncols = min(p.NC, Nh - n0);          // real columns of this group
    const int ntile = (ncols + 15) >> 4;
    const int ntiles_rows = (a.B + LWS_RT - 1) / LWS_RT, G = gridDim.x;
    IG_TDECL;
    IG_T();

    // ---- this thread's float4 slots of a row tile: slot e4 -> (segment, row, k); a float4 is one k-step (k = 4 s + 0..3)
    int srow[LWS_NJ], sdst[LWS_NJ], sk[LWS_NJ], sseg[LWS_NJ];
    bool sok[LWS_NJ];
    const int nvec = LWS_RT * K4 * a.nseg;
#pragma unroll
    for (int j = 0; j < LWS_NJ; ++j) {
        const int e = tid + LWS_THREADS * j;
        sok[j] = e < nvec;
        const int ec = sok[j] ? e : 0;
        const int seg = ec / (LWS_RT * K4), e2 = ec - seg * LWS_RT * K4;
        srow[j] = e2 / K4;
        sk[j] = (e2 - srow[j] * K4) << 2;
        sseg[j] = seg;
        sdst[j] = srow[j] * p.S4 + ((seg * a.K + sk[j]) >> 2);  // + lk * PX per component
    }
    f32x4 xv[LWS_NJ], yv[LWS_NJ];
    const bool hasy = a.A2 != nullptr;
    auto load = [&](int t) __attribute__((always_inline)) {
        const int r0 = t * LWS_RT;
#pragma unroll
        for (int j = 0; j < LWS_NJ; ++j) {
            const int r = min(r0 + srow[j], a.B - 1);
            const float* Ag = sseg[j] ? a.Ab : a.A;
            xv[j] = *reinterpret_cast<const f32x4*>(Ag + (int64_t)r * a.lda + sk[j]);
            if (hasy) {
                const float* Yg = sseg[j] ? a.A2b : a.A2;
                yv[j] = *reinterpret_cast<const f32x4*>(Yg + (int64_t)r * a.lda2 + sk[j]);
            }
        }
    };
    if ((int)blockIdx.x < ntiles_rows) load(blockIdx.x);
    IG_T();

    // ---- both images start as zeroes (the k-steps past Kt / 4, the columns past the real ones)
    for (int e = tid; e < 4 * (p.PW + p.PX); e += LWS_THREADS) smem[e] = 0.f;
    __syncthreads();
    // ---- the column group's weights, once: Wl[k & 3][n - n0][k >> 2] = W'(k, n), k counted over the segments
    {
        const bool nfast = a.wsn == 1;
        for (int seg = 0; seg < a.nseg; ++seg) {
            const float* Wg = seg ? a.Wb : Wh;
            if (nfast) {  // W'[k][n], n contiguous: a float4 is 4 columns of one k
                const int c4 = p.NC >> 2;
                for (int e = tid; e < a.K * c4; e += LWS_THREADS) {
                    const int k = e / c4, n = (e - k * c4) << 2;
                    if (n0 + n < Nh) {  // (N % 4 == 0)
                        const f32x4 v = *reinterpret_cast<const f32x4*>(Wg + (int64_t)k * a.wsk + n0 + n);
                        const int kk = seg * a.K + k;
                        float* d = Wl + (kk & 3) * p.PW + n * p.S4 + (kk >> 2);
                        d[0] = v[0]; d[p.S4] = v[1]; d[2 * p.S4] = v[2]; d[3 * p.S4] = v[3];
                    }
                }
            } else {      // W'[n][k], k contiguous: a float4 is one k-step of one column; a wave instruction takes 16
                          // columns x 4 k-steps
                for (int e = tid; e < p.NC * K4; e += LWS_THREADS) {
                    const int r = e >> 4, nt = r / K4;
                    const int n = nt * 16 + (e & 15), k = (r - nt * K4) << 2;
                    if (n0 + n < Nh) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(Wg + (int64_t)(n0 + n) * a.wsn + k);
                        float* d = Wl + n * p.S4 + ((seg * a.K + k) >> 2);
                        d[0] = v[0]; d[p.PW] = v[1]; d[2 * p.PW] = v[2]; d[3 * p.PW] = v[3];
                    }
                }
            }
        }
    }
    float bias_r[LWS_MAXNT];
#pragma unroll
    for (int i = 0; i < LWS_MAXNT; ++i) {
        const int n = n0 + (wid + LWS_WAVES * i) * 16 + li;
        bias_r[i] = (biash != nullptr && n < Nh) ? biash[n] : 0.f;
    }
    const float* ap = Xl + lk * p.PX + li * p.S4;
    const float* wp = Wl + lk * p.PW + (wid * 16 + li) * p.S4;
    const int ngroups = ((Kt >> 2) + 3) >> 2;  // groups of four k-steps (the last one padded with zeroes)
    const int nmy = wid < ntile ? (ntile - wid + LWS_WAVES - 1) / LWS_WAVES : 0;  // column tiles of this wave
    IG_T();

    for (int t = blockIdx.x; t < ntiles_rows; t += G) {
        IG_T();
        lds_barrier();  // the previous tile's operand reads are done (first round: nothing).  (LDS-only barriers: __syncthreads()
                        // would also wait for the prefetch just issued and the epilogue's stores, a load latency per tile)
#pragma unroll
        for (int j = 0; j < LWS_NJ; ++j) {
            if (!sok[j]) continue;
            f32x4 v = xv[j];
            if (hasy) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] *= act_grad_from_out(yv[j][u], a.a_act, a.a_slope);
            }
            float* d = Xl + sdst[j];
            d[0] = v[0]; d[p.PX] = v[1]; d[2 * p.PX] = v[2]; d[3 * p.PX] = v[3];
        }
        if (t + G < ntiles_rows) load(t + G);
        IG_T();
        lds_barrier();  // the tile (first round: also the weights) is staged
        IG_T();
        f32x4 acc[LWS_MAXNT];
#pragma unroll
        for (int i = 0; i < LWS_MAXNT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (one loop per count of column tiles this wave owns: a test inside the k loop is a branch per MFMA)
        auto kloop = [&](auto cnt) __attribute__((always_inline)) {
            constexpr int NT = decltype(cnt)::value;
            // two register sets: the operand reads of group g + 1 are issued BEFORE the MFMAs of group g (a wave's
            // MFMAs queue on the matrix pipe, and nothing behind them in its instruction stream issues meanwhile)
            f32x4 av[2], bv[2][NT];
            auto fetch = [&](int g, int set) __attribute__((always_inline)) {
                av[set] = *reinterpret_cast<const f32x4*>(ap + 4 * g);
#pragma unroll
                for (int i = 0; i < NT; ++i) bv[set][i] = *reinterpret_cast<const f32x4*>(wp + i * 16 * LWS_WAVES * p.S4 + 4 * g);
            };
            auto fma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][u], bv[set][i][u], acc[i], 0, 0, 0);
            };
            fetch(0, 0);
            int g = 0;
            for (; g + 2 <= ngroups; g += 2) {
                fetch(g + 1, 1);
                fma(0);
                if (g + 2 < ngroups) fetch(g + 2, 0);
                fma(1);
            }
            if (g < ngroups) fma(0);
        };
        if (nmy == 2) kloop(std::integral_constant<int, 2>{});
        else if (nmy == 1) kloop(std::integral_constant<int, 1>{});
        IG_T();
        // D: row = lk*4 + r (sample), column = li
        const int r0 = t * LWS_RT;
#pragma unroll
        for (int i = 0; i < LWS_MAXNT; ++i) {
            const int n = n0 + (wid + LWS_WAVES * i) * 16 + li;
            if (wid + LWS_WAVES * i >= ntile || n >= Nh) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = r0 + lk * 4 + r;
                if (m < a.B) {
                    float v = act_apply(acc[i][r] + bias_r[i], a.act, a.slope);
                    float* cp = Ch + (int64_t)m * ldch + n;
                    if (a.accumulate) v += *cp;  // (uniform)
                    *cp = v;
                }
            }
        }
    }
    IG_TPRINT("lin_ws: load0 | W staging | per tile: (epilogue) [barrier] stage+load [barrier] k-loop ...");
}

// ---- weight / bias gradient ----------------------------------------------------------------
#define LW_BB 64   // samples per staged chunk
#define LW_BN 64   // output rows (n) per workgroup: one 16-row tile per wave, each against the four 16-column k-tiles
#define LW_YS 80   // row strides == 16 (mod 32): the four k-rows of an MFMA step hit disjoint banks
#define LW_XS 80

struct LinWgArgs {
    const float* dY;  // [B][N] rows lddy
    const float* Y;   // [B][N] rows ldy (for act'), may be null
    const float* X;   // [B][K] rows ldx
    float* dW;        // [N][K]
    float* db;        // [N] or null
    float* part;      // [gx][N*K + N] per-slice partials (the op's r[10] region)
    int lddy, ldy, ldx;
    int B, N, K;
    int act;
    float slope;
    int chunks_per_block;  // consecutive LW_BB-sample chunks handled by one workgroup
    int gx, gy, gz;        // workgroups of this layer: batch slices x 64-row tiles x 64-column tiles
    int first_block;       // of this layer in the flattened grid
    int vec;               // rows of dY, Y and X are 16-byte aligned and N, K multiples of 4: dwordx4 staging
};

// The weight gradients of ALL Linear layers of a backward pass in one launch (they only have to be done
// before the optimiser; each is a few microseconds of latency-bound work, so running them side by side
// costs about as much as running one).  Flat 1-D grid; a workgroup finds its layer from first_block.
#define LIN_WG_MAX_LAYERS 8
struct LinWgMulti {
    LinWgArgs l[LIN_WG_MAX_LAYERS];
    int n;
};

// One workgroup's tile.  VEC (the layer's a.vec) is a template argument, not a run-time branch inside the staging code:
// with both shapes of the prefetch registers in one body the compiler left three of them in scratch memory, and every
// access to those drained the next chunk's loads (scratch counts in vmcnt).
template <bool BF, bool VEC>
__device__ __forceinline__ void lw_tile(const LinWgArgs& a, float* Ys, float* Xs) {
    const int bid = blockIdx.x - a.first_block;
    const int bx = bid % a.gx, by = (bid / a.gx) % a.gy, bz = bid / (a.gx * a.gy);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int n0 = by * LW_BN, k0 = bz * 64;
    const int nt = wid;  // this wave's 16 output rows; it keeps all four k-tiles (a 64 x 64 block tile: X is read once
                         // per 64 output rows and dY once per 64 columns, 3/4 of the traffic of 32 x 64)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // Chunks are software-pipelined: the global loads of chunk i+1 are issued (into registers) before the
    // MFMA loop of chunk i and only waited for when they are written to LDS one iteration later.
    const int c_first = bx * a.chunks_per_block;
    const int c_end = min(c_first + a.chunks_per_block, (a.B + LW_BB - 1) / LW_BB);
    const bool has_act = a.act != VAE_ACT_NONE;
    float vy[16], vyy[16], vx[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) vyy[j] = 0.f;
    constexpr bool vec = VEC;
    auto load_chunk = [&](int ch) __attribute__((always_inline)) {  // 64 rows x 64 dY (and Y) columns, 64 rows x 64 X columns; zero past the edges
        const int b0 = ch * LW_BB;
        if constexpr (vec) {
            // four float4 per tensor and thread, every load unconditional with a clamped address (a load under a branch
            // is waited for on the spot); what lies past an edge is zeroed when it goes to LDS
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = tid + j * LIN_THREADS;  // 1024 = 64 rows x 16 float4
                const int r = min(b0 + (e >> 4), a.B - 1), cn = min(n0 + ((e & 15) << 2), a.N - 4), ck = min(k0 + ((e & 15) << 2), a.K - 4);
                const float4 y4 = *reinterpret_cast<const float4*>(a.dY + (int64_t)r * a.lddy + cn);
                const float4 x4 = *reinterpret_cast<const float4*>(a.X + (int64_t)r * a.ldx + ck);
                vy[4 * j] = y4.x; vy[4 * j + 1] = y4.y; vy[4 * j + 2] = y4.z; vy[4 * j + 3] = y4.w;
                vx[4 * j] = x4.x; vx[4 * j + 1] = x4.y; vx[4 * j + 2] = x4.z; vx[4 * j + 3] = x4.w;
                if (has_act) {
                    const float4 q4 = *reinterpret_cast<const float4*>(a.Y + (int64_t)r * a.ldy + cn);
                    vyy[4 * j] = q4.x; vyy[4 * j + 1] = q4.y; vyy[4 * j + 2] = q4.z; vyy[4 * j + 3] = q4.w;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = tid + j * LIN_THREADS;  // 4096 = 64 x 64
            const int r = e >> 6, c = e & 63;
            vy[j] = 0.f;
            vyy[j] = 0.f;
            if (b0 + r < a.B && n0 + c < a.N) {
                vy[j] = a.dY[(int64_t)(b0 + r) * a.lddy + n0 + c];
                if (has_act) vyy[j] = a.Y[(int64_t)(b0 + r) * a.ldy + n0 + c];
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = tid + j * LIN_THREADS;  // 4096 = 64 x 64
            const int r = e >> 6, c = e & 63;
            vx[j] = (b0 + r < a.B && k0 + c < a.K) ? a.X[(int64_t)(b0 + r) * a.ldx + k0 + c] : 0.f;
        }
    };
    if (c_first < c_end) load_chunk(c_first);
    for (int ch = c_first; ch < c_end; ++ch) {
        __syncthreads();
        if constexpr (vec) {
            const int b0 = ch * LW_BB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = tid + j * LIN_THREADS, r = e >> 4, c4 = (e & 15) << 2;
                const bool rok = b0 + r < a.B, yok = rok && n0 + c4 < a.N, xok = rok && k0 + c4 < a.K;
                float4 y4, x4;
                y4.x = yok ? (has_act ? vy[4 * j] * act_grad_from_out(vyy[4 * j], a.act, a.slope) : vy[4 * j]) : 0.f;
                y4.y = yok ? (has_act ? vy[4 * j + 1] * act_grad_from_out(vyy[4 * j + 1], a.act, a.slope) : vy[4 * j + 1]) : 0.f;
                y4.z = yok ? (has_act ? vy[4 * j + 2] * act_grad_from_out(vyy[4 * j + 2], a.act, a.slope) : vy[4 * j + 2]) : 0.f;
                y4.w = yok ? (has_act ? vy[4 * j + 3] * act_grad_from_out(vyy[4 * j + 3], a.act, a.slope) : vy[4 * j + 3]) : 0.f;
                x4.x = xok ? vx[4 * j] : 0.f; x4.y = xok ? vx[4 * j + 1] : 0.f; x4.z = xok ? vx[4 * j + 2] : 0.f; x4.w = xok ? vx[4 * j + 3] : 0.f;
                *reinterpret_cast<float4*>(Ys + r * LW_YS + c4) = y4;
                *reinterpret_cast<float4*>(Xs + r * LW_XS + c4) = x4;
            }
        } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = tid + j * LIN_THREADS;
            Ys[(e >> 6) * LW_YS + (e & 63)] = has_act ? vy[j] * act_grad_from_out(vyy[j], a.act, a.slope) : vy[j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = tid + j * LIN_THREADS;
            Xs[(e >> 6) * LW_XS + (e & 63)] = vx[j];
        }
        }
        __syncthreads();
        if (ch + 1 < c_end) load_chunk(ch + 1);
        if (a.db != nullptr && bz == 0 && tid < LW_BN) {
            float t = 0.f;
            for (int r = 0; r < LW_BB; ++r) t += Ys[r * LW_YS + tid];
            bsum += t;
        }
        if constexpr (BF) {  // sixteen samples per bf16 MFMA: this lane's are b16 + lk*4 + 0..3
            const float* ypb = Ys + lk * 4 * LW_YS + nt * 16 + li;
            const float* xpb = Xs + lk * 4 * LW_XS + li;
#pragma unroll
            for (int b16 = 0; b16 < LW_BB; b16 += 16) {
                const s16x4 av = bf16_pack4(ypb[b16 * LW_YS], ypb[(b16 + 1) * LW_YS], ypb[(b16 + 2) * LW_YS], ypb[(b16 + 3) * LW_YS]);
                auto xq = [&](int c) {
                    return bf16_pack4(xpb[b16 * LW_XS + c], xpb[(b16 + 1) * LW_XS + c], xpb[(b16 + 2) * LW_XS + c],
                                      xpb[(b16 + 3) * LW_XS + c]);
                };
                acc0 = MFMA_BF16(av, xq(0), acc0);
                acc1 = MFMA_BF16(av, xq(16), acc1);
                acc2 = MFMA_BF16(av, xq(32), acc2);
                acc3 = MFMA_BF16(av, xq(48), acc3);
            }
        } else {
        const float* yp = Ys + lk * LW_YS + nt * 16 + li;
        const float* xp = Xs + lk * LW_XS + li;
#pragma unroll 4
        for (int b4 = 0; b4 < LW_BB; b4 += 4) {
            const float av = yp[b4 * LW_YS];
            const float x0 = xp[b4 * LW_XS], x1 = xp[b4 * LW_XS + 16], x2 = xp[b4 * LW_XS + 32], x3 = xp[b4 * LW_XS + 48];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x2, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, x3, acc3, 0, 0, 0);
        }
        }
    }
    float* const prow = a.part + (int64_t)bx * ((int64_t)a.N * a.K + a.N);
    if (a.db != nullptr && bz == 0 && tid < LW_BN && n0 + tid < a.N) prow[(int64_t)a.N * a.K + n0 + tid] = bsum;
    // D: row (n) = lk*4 + r, col (k) = li
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + nt * 16 + lk * 4 + r;
        const int k = k0 + li;
        if (n < a.N) {
            if (k < a.K) prow[(int64_t)n * a.K + k] = acc0[r];
            if (k + 16 < a.K) prow[(int64_t)n * a.K + k + 16] = acc1[r];
            if (k + 32 < a.K) prow[(int64_t)n * a.K + k + 32] = acc2[r];
            if (k + 48 < a.K) prow[(int64_t)n * a.K + k + 48] = acc3[r];
        }
    }
}

template <bool BF>
__global__ __launch_bounds__(LIN_THREADS) void lin_wgrad_k(LinWgMulti m) {
    __shared__ __attribute__((aligned(16))) float Ys[LW_BB * LW_YS];  // dY' chunk: [b][n0..n0+64)
    __shared__ __attribute__((aligned(16))) float Xs[LW_BB * LW_XS];  // X chunk:   [b][k0..k0+64)
    int layer = 0;
#pragma unroll
    for (int i = 1; i < LIN_WG_MAX_LAYERS; ++i)
        if (i < m.n && (int)blockIdx.x >= m.l[i].first_block) layer = i;
    const LinWgArgs& a = m.l[layer];
    if (a.vec != 0) lw_tile<BF, true>(a, Ys, Xs);
    else lw_tile<BF, false>(a, Ys, Xs);
}

// ---- weight / bias gradient, whole layer per workgroup (round 3) ----------------------------------------------------
// lin_wgrad_k gives a workgroup a 64 x 64 tile of dW: X is re-read once per 64 output rows and dY once per 64
// columns (129 MB through L2 for 25 MB of activations at B = 4096).  Here ONE workgroup of 16 waves owns the whole
// dW[N][K] of a layer for its batch slice - up to 32 x 7 accumulator tiles spread over the waves, at most 14 per wave -
// so every row of dY, Y and X is read exactly once: a 64-row chunk goes to LDS as it lies in memory ([b][n] and [b][k]
// are the MFMA's A and B operands with the batch as the reduction index), 16 k-steps run on it, next chunk.  The
// larger of the two tile counts is split over the waves (tiles w, w + 16), the smaller (<= 7) is walked by every wave.
// Partial rows [slice][N*K + N] as before; wgrad_finalize_k sums them in a fixed order.
#define LW2_THREADS 1024
#define LW2_WAVES 16
#define LW2_RB 64     // == LW_BB: the slices of both kernels are whole 64-row chunks
#define LW2_FULLMAX 7
__host__ __device__ inline int lw2_stride(int cols) {  // >= cols, = 16 (mod 32): the four batch rows of an MFMA step hit disjoint banks
    int s = (cols + 15) & ~15;
    if ((s & 31) != 16) s += 16;
    return s;
}
__host__ __device__ inline int lw2_fullc(int t) { return t <= 1 ? 1 : t <= 2 ? 2 : t <= 4 ? 4 : LW2_FULLMAX; }

template <bool SPLIT_N, int TFC>
__device__ __forceinline__ void lw2_body(const LinWgArgs& a, int slice, float* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int TN = (a.N + 15) >> 4, TK = (a.K + 15) >> 4;
    const int NS = lw2_stride(SPLIT_N ? a.N : 16 * TFC), KS = lw2_stride(SPLIT_N ? 16 * TFC : a.K);
    float* Yl = smem;                 // dY' chunk [64][NS]
    float* Xl = smem + LW2_RB * NS;   // X chunk   [64][KS]
    const int TS = SPLIT_N ? TN : TK;
    const int nsp = wid < TS ? (TS - wid + LW2_WAVES - 1) / LW2_WAVES : 0;  // split-dimension tiles of this wave (<= 2)
    f32x4 acc[2][TFC];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TFC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool has_act = a.act != VAE_ACT_NONE, vec = a.vec != 0;
    const int nchunks = (a.B + LW2_RB - 1) / LW2_RB;
    const int c_first = slice * a.chunks_per_block, c_end = min(c_first + a.chunks_per_block, nchunks);
    // the pad columns (and whatever a TFC wider than the layer reads) stay zero: staging writes real columns only
    for (int e = tid; e < LW2_RB * (NS + KS); e += LW2_THREADS) smem[e] = 0.f;

    for (int ch = c_first; ch < c_end; ++ch) {
        const int b0 = ch * LW2_RB;
        __syncthreads();  // the zeroes / the previous chunk's operand reads are done
        if (vec) {
            auto stage = [&](const float* src, const float* src2, int ld, int ld2, int cols, float* dst, int stride) __attribute__((always_inline)) {
                const int c4 = cols >> 2, total = LW2_RB * c4;
                for (int base = tid; base < total; base += 4 * LW2_THREADS) {
                    f32x4 v[4], y[4];
                    int rr[4], cc[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {  // every load unconditional with a clamped address
                        const int e = min(base + j * LW2_THREADS, total - 1);
                        rr[j] = e / c4;
                        cc[j] = (e - rr[j] * c4) << 2;
                        const int gr = min(b0 + rr[j], a.B - 1);
                        v[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)gr * ld + cc[j]);
                        if (src2 != nullptr) y[j] = *reinterpret_cast<const f32x4*>(src2 + (int64_t)gr * ld2 + cc[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (base + j * LW2_THREADS >= total) continue;
                        f32x4 o = v[j];
                        if (src2 != nullptr) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) o[u] *= act_grad_from_out(y[j][u], a.act, a.slope);
                        }
                        if (b0 + rr[j] >= a.B) o = f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the batch add nothing
                        *reinterpret_cast<f32x4*>(dst + rr[j] * stride + cc[j]) = o;
                    }
                }
            };
            stage(a.dY, has_act ? a.Y : nullptr, a.lddy, a.ldy, a.N, Yl, NS);
            stage(a.X, nullptr, a.ldx, 0, a.K, Xl, KS);
        } else {
            for (int e = tid; e < LW2_RB * a.N; e += LW2_THREADS) {
                const int r = e / a.N, c = e - r * a.N;
                float v = 0.f;
                if (b0 + r < a.B) {
                    v = a.dY[(int64_t)(b0 + r) * a.lddy + c];
                    if (has_act) v *= act_grad_from_out(a.Y[(int64_t)(b0 + r) * a.ldy + c], a.act, a.slope);
                }
                Yl[r * NS + c] = v;
            }
            for (int e = tid; e < LW2_RB * a.K; e += LW2_THREADS) {
                const int r = e / a.K, c = e - r * a.K;
                Xl[r * KS + c] = b0 + r < a.B ? a.X[(int64_t)(b0 + r) * a.ldx + c] : 0.f;
            }
        }
        __syncthreads();
        if (a.db != nullptr && tid < a.N) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < LW2_RB; ++r) t += Yl[r * NS + tid];
            bsum += t;
        }
        // D[n][k] += sum_b dY'[b][n] * X[b][k]: A = Yl (row n = li, reduction index b = lk), B = Xl (b = lk, column k = li)
        auto kloop = [&](auto cnt) __attribute__((always_inline)) {
            constexpr int NSP = decltype(cnt)::value;
            const float* sp = (SPLIT_N ? Yl : Xl) + lk * (SPLIT_N ? NS : KS) + wid * 16 + li;  // split operand, tile wid (+16 tiles)
            const float* fp = (SPLIT_N ? Xl : Yl) + lk * (SPLIT_N ? KS : NS) + li;             // full operand, tiles 0..TFC-1
            const int sst = 4 * (SPLIT_N ? NS : KS), fst = 4 * (SPLIT_N ? KS : NS);
#pragma unroll 2
            for (int s = 0; s < LW2_RB / 4; ++s) {
                float sv[NSP], fv[TFC];
#pragma unroll
                for (int i = 0; i < NSP; ++i) sv[i] = sp[s * sst + i * 16 * LW2_WAVES];
#pragma unroll
                for (int j = 0; j < TFC; ++j) fv[j] = fp[s * fst + j * 16];
#pragma unroll
                for (int i = 0; i < NSP; ++i)
#pragma unroll
                    for (int j = 0; j < TFC; ++j)
                        acc[i][j] = SPLIT_N ? __builtin_amdgcn_mfma_f32_16x16x4f32(sv[i], fv[j], acc[i][j], 0, 0, 0)
                                            : __builtin_amdgcn_mfma_f32_16x16x4f32(fv[j], sv[i], acc[i][j], 0, 0, 0);
            }
        };
        if (nsp == 2) kloop(std::integral_constant<int, 2>{});
        else if (nsp == 1) kloop(std::integral_constant<int, 1>{});
    }
    float* const prow = a.part + (int64_t)slice * ((int64_t)a.N * a.K + a.N);
    if (a.db != nullptr && tid < a.N) prow[(int64_t)a.N * a.K + tid] = bsum;
    // D: row (n) = lk*4 + r, column (k) = li
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i >= nsp) continue;
#pragma unroll
        for (int j = 0; j < TFC; ++j) {
            const int nt = SPLIT_N ? wid + i * LW2_WAVES : j, kt = SPLIT_N ? j : wid + i * LW2_WAVES;
            const int k = kt * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nt * 16 + lk * 4 + r;
                if (n < a.N && k < a.K) prow[(int64_t)n * a.K + k] = acc[i][j][r];
            }
        }
    }
}

__global__ __launch_bounds__(LW2_THREADS) void lin_wgrad2_k(LinWgMulti m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int layer = 0;
#pragma unroll
    for (int i = 1; i < LIN_WG_MAX_LAYERS; ++i)
        if (i < m.n && (int)blockIdx.x >= m.l[i].first_block) layer = i;
    const LinWgArgs& a = m.l[layer];
    const int slice = blockIdx.x - a.first_block;
    const int TN = (a.N + 15) >> 4, TK = (a.K + 15) >> 4;
    const bool split_n = TN >= TK;
    const int tfc = lw2_fullc(split_n ? TK : TN);
#define LW2_CASE(SN, T) lw2_body<SN, T>(a, slice, smem)
    if (split_n) {
        if (tfc == 1) LW2_CASE(true, 1); else if (tfc == 2) LW2_CASE(true, 2); else if (tfc == 4) LW2_CASE(true, 4); else LW2_CASE(true, LW2_FULLMAX);
    } else {
        if (tfc == 1) LW2_CASE(false, 1); else if (tfc == 2) LW2_CASE(false, 2); else if (tfc == 4) LW2_CASE(false, 4); else LW2_CASE(false, LW2_FULLMAX);
    }
#undef LW2_CASE
}
