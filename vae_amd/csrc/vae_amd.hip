// vae_amd.hip - program executor and C ABI of libvae_amd.so (see include/vae_amd.h).
//
// Host code here only validates shapes/extents and enqueues kernels on the caller's stream:
// no allocation, no synchronisation, so any call sequence can be captured into a hipGraph
// (vae_graph_*), which is how the Python host removes per-kernel launch overhead from the
// train step.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "chan.h"
#include "common.h"
#include "conv.h"
#include "elem.h"
#include "gemm.h"

static thread_local char g_err[512] = "";
void vae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vae_last_error(void) { return g_err; }

extern "C" int vae_check_device(void) {
    int dev = 0;
    VAE_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    VAE_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        VAE_FAIL(VAE_ENODEV, "device %d is %s, libvae_amd is built for gfx950 only", dev, prop.gcnArchName);
    return VAE_OK;
}

// ---- reference resolution ------------------------------------------------------------------
struct Ctx {
    float* const* bases;
    const int64_t* sizes;
    int n_bases;
    int B;
    hipStream_t s;
};
static inline int ref_space(int64_t r) { return (int)(r >> 56); }
static inline int64_t ref_off(int64_t r) { return r & (((int64_t)1 << 56) - 1); }

// Resolves a ref whose extent is `count` floats; nullptr + error when absent/out of bounds.
static float* resolve(const Ctx& c, int64_t ref, int64_t count, const char* what, bool* ok) {
    if (ref < 0) {
        vae_set_error("%s: missing buffer reference", what);
        *ok = false;
        return nullptr;
    }
    const int sp = ref_space(ref);
    const int64_t off = ref_off(ref);
    if (sp >= c.n_bases || c.bases[sp] == nullptr) {
        vae_set_error("%s: address space %d not bound", what, sp);
        *ok = false;
        return nullptr;
    }
    if (count < 0 || off + count > c.sizes[sp]) {
        vae_set_error("%s: extent [%lld, %lld) exceeds space %d of %lld floats", what, (long long)off,
                      (long long)(off + count), sp, (long long)c.sizes[sp]);
        *ok = false;
        return nullptr;
    }
    return c.bases[sp] + off;
}
static float* resolve_opt(const Ctx& c, int64_t ref, int64_t count, const char* what, bool* ok) {
    return ref < 0 ? nullptr : resolve(c, ref, count, what, ok);
}

static inline int ew_grid(int64_t n) {
    int64_t g = ceil_div64(n, 256);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}
// batch chunking for the two-stage channel reductions
static inline void chunking(int B, int64_t per_sample, int* bchunk, int* nchunks) {
    // aim for >= ~16k elements per block, at most VAE_MAX_CHUNKS chunks
    int64_t want = ceil_div64((int64_t)B * per_sample, 16384);
    if (want > VAE_MAX_CHUNKS) want = VAE_MAX_CHUNKS;
    if (want < 1) want = 1;
    if (want > B) want = B;
    *bchunk = (int)ceil_div64(B, want);
    *nchunks = (int)ceil_div64(B, *bchunk);
}

// ---- conv launches ---------------------------------------------------------------------------
static ConvGeom geom_of(const vae_op& op) {
    ConvGeom g;
    g.Cin = op.d[0]; g.H = op.d[1]; g.W = op.d[2]; g.Cout = op.d[3]; g.OH = op.d[4]; g.OW = op.d[5];
    g.KH = op.d[6]; g.KW = op.d[7]; g.sh = op.d[8]; g.sw = op.d[9]; g.ph = op.d[10]; g.pw = op.d[11];
    return g;
}
static int check_geom(const vae_op& op) {
    const ConvGeom g = geom_of(op);
    if (g.Cin < 1 || g.Cout < 1 || g.H < 1 || g.W < 1 || g.OH < 1 || g.OW < 1 || g.KH < 1 || g.KW < 1 ||
        g.sh < 1 || g.sw < 1 || g.ph < 0 || g.pw < 0)
        VAE_FAIL(VAE_EINVAL, "conv: non-positive geometry");
    if (op.kind == VAE_OP_CONV) {
        if (g.OH != (g.H + 2 * g.ph - g.KH) / g.sh + 1 || g.OW != (g.W + 2 * g.pw - g.KW) / g.sw + 1)
            VAE_FAIL(VAE_EINVAL, "conv: output %dx%d inconsistent with input %dx%d k%dx%d s%d,%d p%d,%d", g.OH,
                     g.OW, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
    } else {
        const int oph = g.OH - ((g.H - 1) * g.sh - 2 * g.ph + g.KH), opw = g.OW - ((g.W - 1) * g.sw - 2 * g.pw + g.KW);
        if (oph < 0 || oph >= g.sh || opw < 0 || opw >= g.sw)
            VAE_FAIL(VAE_EINVAL, "convT: output %dx%d inconsistent with input %dx%d k%dx%d s%d,%d p%d,%d", g.OH,
                     g.OW, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
    }
    return VAE_OK;
}

// y = conv(x): geometry in "conv orientation" (Cin,H,W) -> (Cout,OH,OW)
static int launch_conv_gather(const Ctx& c, const float* x, const float* w, const float* bias, float* y,
                              const ConvGeom& g, int act, float slope, int accumulate) {
    const int64_t npix = (int64_t)c.B * g.OH * g.OW;
    const int gx = (int)ceil_div64(npix, 256);
    if (g.Cout > 4) {
        dim3 grid(gx, (g.Cout + 7) / 8);
        hipLaunchKernelGGL(conv_gather_k<8>, grid, dim3(256), 0, c.s, x, w, bias, y, c.B, g, act, slope, accumulate);
    } else {
        dim3 grid(gx, (g.Cout + 3) / 4);
        hipLaunchKernelGGL(conv_gather_k<4>, grid, dim3(256), 0, c.s, x, w, bias, y, c.B, g, act, slope, accumulate);
    }
    VAE_LAUNCH_CHECK("conv_gather_k");
    return VAE_OK;
}
// y = convT(x): geometry in "convT orientation" (Cin,H,W) -> (Cout,OH,OW), OH ~ H*sh
static int launch_convT_gather(const Ctx& c, const float* x, const float* w, const float* bias, float* y,
                               const ConvGeom& g, int act, float slope, int accumulate) {
    const int AH = (g.OH + g.sh - 1) / g.sh, AW = (g.OW + g.sw - 1) / g.sw;
    const int64_t npix = (int64_t)c.B * AH * AW;
    const int gx = (int)ceil_div64(npix, 256);
    if (g.Cout > 4) {
        dim3 grid(gx, (g.Cout + 7) / 8, g.sh * g.sw);
        hipLaunchKernelGGL(convT_gather_k<8>, grid, dim3(256), 0, c.s, x, w, bias, y, c.B, g, act, slope, accumulate);
    } else {
        dim3 grid(gx, (g.Cout + 3) / 4, g.sh * g.sw);
        hipLaunchKernelGGL(convT_gather_k<4>, grid, dim3(256), 0, c.s, x, w, bias, y, c.B, g, act, slope, accumulate);
    }
    VAE_LAUNCH_CHECK("convT_gather_k");
    return VAE_OK;
}
// dw[a][cc][kh][kw] += sum S[b,a,oh,ow] * G[b,cc,oh*sh-ph+kh,...]
static int launch_wgrad(const Ctx& c, const float* S, const float* G, float* dw, int A, int SH, int SW, int C,
                        int GH, int GW, int KH, int KW, int sh, int sw, int ph, int pw) {
    const int64_t R = (int64_t)c.B * SH * SW;
    const int AT = 4, TG = (KH * KW <= 9) ? 9 : 16;
    const int tiles = ((A + AT - 1) / AT) * C;
    const int zg = (KH * KW + TG - 1) / TG;
    int64_t want_chunks = ceil_div64(2048, (int64_t)tiles * zg);
    if (want_chunks < 1) want_chunks = 1;
    int64_t chunk = ceil_div64(R, want_chunks);
    if (chunk < 2048) chunk = 2048;  // >= 8 pixels per lane before paying the block reduction
    const int gx = (int)ceil_div64(R, chunk);
    dim3 grid(gx, tiles, zg);
    if (TG == 9)
        hipLaunchKernelGGL((conv_wgrad_k<4, 9>), grid, dim3(256), 0, c.s, S, G, dw, c.B, A, SH, SW, C, GH, GW, KH, KW,
                           sh, sw, ph, pw, chunk);
    else
        hipLaunchKernelGGL((conv_wgrad_k<4, 16>), grid, dim3(256), 0, c.s, S, G, dw, c.B, A, SH, SW, C, GH, GW, KH,
                           KW, sh, sw, ph, pw, chunk);
    VAE_LAUNCH_CHECK("conv_wgrad_k");
    return VAE_OK;
}

static int conv_forward(const Ctx& c, const vae_op& op) {
    int rc = check_geom(op);
    if (rc) return rc;
    const ConvGeom g = geom_of(op);
    bool ok = true;
    const int64_t nin = (int64_t)c.B * g.Cin * g.H * g.W, nout = (int64_t)c.B * g.Cout * g.OH * g.OW;
    const int64_t nw = (int64_t)g.Cin * g.Cout * g.KH * g.KW;
    const float* x = resolve(c, op.r[0], nin, "conv.in", &ok);
    float* y = resolve(c, op.r[1], nout, "conv.out", &ok);
    const float* w = resolve(c, op.r[4], nw, "conv.w", &ok);
    const float* b = resolve_opt(c, op.r[5], g.Cout, "conv.b", &ok);
    if (!ok) return VAE_EINVAL;
    if (op.kind == VAE_OP_CONV) return launch_conv_gather(c, x, w, b, y, g, op.act, op.f[0], 0);
    return launch_convT_gather(c, x, w, b, y, g, op.act, op.f[0], 0);
}

static int conv_backward(const Ctx& c, const vae_op& op) {
    int rc = check_geom(op);
    if (rc) return rc;
    const ConvGeom g = geom_of(op);
    bool ok = true;
    const int64_t nin = (int64_t)c.B * g.Cin * g.H * g.W, nout = (int64_t)c.B * g.Cout * g.OH * g.OW;
    const int64_t nw = (int64_t)g.Cin * g.Cout * g.KH * g.KW;
    int bchunk, nchunks;
    chunking(c.B, (int64_t)g.OH * g.OW, &bchunk, &nchunks);
    const float* x = resolve(c, op.r[0], nin, "conv.in", &ok);
    const float* y = resolve(c, op.r[1], nout, "conv.out", &ok);
    float* dx = resolve_opt(c, op.r[2], nin, "conv.din", &ok);
    float* dy = resolve(c, op.r[3], nout, "conv.dout", &ok);
    const float* w = resolve(c, op.r[4], nw, "conv.w", &ok);
    // gradients live at the same offsets in the GRAD space (space 2)
    float* dw = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[4]), nw, "conv.dw", &ok);
    float* db = op.r[5] < 0 ? nullptr : resolve(c, ((int64_t)2 << 56) | ref_off(op.r[5]), g.Cout, "conv.db", &ok);
    float* part = resolve(c, op.r[6], (int64_t)2 * g.Cout * VAE_MAX_CHUNKS, "conv.partials", &ok);
    if (!ok) return VAE_EINVAL;

    // 1) dy <- dy * act'(y) in place, bias gradient = per-channel sum
    if (op.act != VAE_ACT_NONE || db != nullptr) {
        hipLaunchKernelGGL(chan_reduce_k<CR_ACTBWD>, dim3(g.Cout, nchunks), dim3(256), 0, c.s, nullptr, dy, y, nullptr,
                           part, c.B, g.Cout, g.OH * g.OW, bchunk, op.act, op.f[0], 0);
        VAE_LAUNCH_CHECK("chan_reduce_k<ACTBWD>");
        if (db != nullptr) {
            hipLaunchKernelGGL(bias_grad_finalize_k, dim3((g.Cout + 63) / 64), dim3(64), 0, c.s, part, db, g.Cout,
                               nchunks);
            VAE_LAUNCH_CHECK("bias_grad_finalize_k");
        }
    }
    // 2) weight gradient, 3) input gradient
    if (op.kind == VAE_OP_CONV) {
        rc = launch_wgrad(c, dy, x, dw, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
        if (rc) return rc;
        if (dx != nullptr) {
            ConvGeom t = g;  // transposed-conv orientation: (Cout,OH,OW) -> (Cin,H,W)
            t.Cin = g.Cout; t.H = g.OH; t.W = g.OW; t.Cout = g.Cin; t.OH = g.H; t.OW = g.W;
            rc = launch_convT_gather(c, dy, w, nullptr, dx, t, VAE_ACT_NONE, 0.f, op.flags & VAE_F_ACC_DIN);
        }
    } else {
        rc = launch_wgrad(c, x, dy, dw, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
        if (rc) return rc;
        if (dx != nullptr) {
            ConvGeom t = g;  // conv orientation: (Cout,OH,OW) -> (Cin,H,W)
            t.Cin = g.Cout; t.H = g.OH; t.W = g.OW; t.Cout = g.Cin; t.OH = g.H; t.OW = g.W;
            rc = launch_conv_gather(c, dy, w, nullptr, dx, t, VAE_ACT_NONE, 0.f, op.flags & VAE_F_ACC_DIN);
        }
    }
    return rc;
}

// ---- batch norm -----------------------------------------------------------------------------
static int bn_forward(const Ctx& c, const vae_op& op, int training) {
    const int C = op.d[0], HW = op.d[1];
    if (C < 1 || HW < 1) VAE_FAIL(VAE_EINVAL, "bn: bad shape C=%d HW=%d", C, HW);
    const int64_t total = (int64_t)c.B * C * HW;
    if (training && (int64_t)c.B * HW < 2)
        VAE_FAIL(VAE_EINVAL, "bn: training needs more than 1 value per channel (got B*HW=%lld)", (long long)c.B * HW);
    bool ok = true;
    const float* x = resolve(c, op.r[0], total, "bn.in", &ok);
    float* y = resolve(c, op.r[1], total, "bn.out", &ok);
    const float* gamma = resolve(c, op.r[4], C, "bn.gamma", &ok);
    const float* beta = resolve(c, op.r[5], C, "bn.beta", &ok);
    float* save = resolve(c, op.r[6], (int64_t)4 * C, "bn.save", &ok);
    float* rm = resolve(c, op.r[7], C, "bn.running_mean", &ok);
    float* rv = resolve(c, op.r[8], C, "bn.running_var", &ok);
    float* part = resolve(c, op.r[9], (int64_t)2 * C * VAE_MAX_CHUNKS, "bn.partials", &ok);
    if (!ok) return VAE_EINVAL;
    if (training) {
        int bchunk, nchunks;
        chunking(c.B, HW, &bchunk, &nchunks);
        hipLaunchKernelGGL(chan_reduce_k<CR_STATS>, dim3(C, nchunks), dim3(256), 0, c.s, x, nullptr, nullptr, nullptr,
                           part, c.B, C, HW, bchunk, 0, 0.f, 0);
        VAE_LAUNCH_CHECK("chan_reduce_k<STATS>");
        hipLaunchKernelGGL(bn_stats_finalize_k, dim3((C + 63) / 64), dim3(64), 0, c.s, x, part, save, rm, rv, C, HW,
                           nchunks, (int64_t)c.B * HW, op.f[0], op.f[1]);
        VAE_LAUNCH_CHECK("bn_stats_finalize_k");
    } else {
        hipLaunchKernelGGL(bn_eval_prepare_k, dim3((C + 63) / 64), dim3(64), 0, c.s, rm, rv, save, C, op.f[0]);
        VAE_LAUNCH_CHECK("bn_eval_prepare_k");
    }
    hipLaunchKernelGGL(bn_apply_k, dim3(ew_grid(total)), dim3(256), 0, c.s, x, y, save, gamma, beta, total, C, HW,
                       (op.flags & VAE_F_BN_RELU) ? 1 : 0);
    VAE_LAUNCH_CHECK("bn_apply_k");
    return VAE_OK;
}

static int bn_backward(const Ctx& c, const vae_op& op) {
    const int C = op.d[0], HW = op.d[1];
    const int64_t total = (int64_t)c.B * C * HW;
    bool ok = true;
    const float* x = resolve(c, op.r[0], total, "bn.in", &ok);
    const float* y = resolve(c, op.r[1], total, "bn.out", &ok);
    float* dx = resolve(c, op.r[2], total, "bn.din", &ok);
    float* dy = resolve(c, op.r[3], total, "bn.dout", &ok);
    const float* gamma = resolve(c, op.r[4], C, "bn.gamma", &ok);
    float* dgamma = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[4]), C, "bn.dgamma", &ok);
    float* dbeta = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[5]), C, "bn.dbeta", &ok);
    float* save = resolve(c, op.r[6], (int64_t)4 * C, "bn.save", &ok);
    float* part = resolve(c, op.r[9], (int64_t)2 * C * VAE_MAX_CHUNKS, "bn.partials", &ok);
    if (!ok) return VAE_EINVAL;
    int bchunk, nchunks;
    chunking(c.B, HW, &bchunk, &nchunks);
    hipLaunchKernelGGL(chan_reduce_k<CR_BNBWD>, dim3(C, nchunks), dim3(256), 0, c.s, x, dy, y, save, part, c.B, C, HW,
                       bchunk, 0, 0.f, (op.flags & VAE_F_BN_RELU) ? 1 : 0);
    VAE_LAUNCH_CHECK("chan_reduce_k<BNBWD>");
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((C + 63) / 64), dim3(64), 0, c.s, part, save, dgamma, dbeta, C, nchunks,
                       (int64_t)c.B * HW);
    VAE_LAUNCH_CHECK("bn_bwd_finalize_k");
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_grid(total)), dim3(256), 0, c.s, x, dy, dx, save, gamma, total, C, HW);
    VAE_LAUNCH_CHECK("bn_bwd_apply_k");
    return VAE_OK;
}

// ---- linear ---------------------------------------------------------------------------------
static int launch_gemm(const Ctx& c, GemmArgs a, int ksplit) {
    if (a.M < 1 || a.N < 1 || a.K < 1) VAE_FAIL(VAE_EINVAL, "gemm: bad shape %dx%dx%d", a.M, a.N, a.K);
    if (ksplit < 1) ksplit = 1;
    a.kchunk = (int)ceil_div64(ceil_div64(a.K, ksplit), GEMM_BK) * GEMM_BK;
    const int gz = (int)ceil_div64(a.K, a.kchunk);
    if (gz > 1 && a.mode != GEMM_ATOMIC) VAE_FAIL(VAE_EINVAL, "gemm: split-K needs atomic mode");
    dim3 grid((a.N + GEMM_BN - 1) / GEMM_BN, (a.M + GEMM_BM - 1) / GEMM_BM, gz);
    hipLaunchKernelGGL(gemm_f32_mfma_k, grid, dim3(256), 0, c.s, a);
    VAE_LAUNCH_CHECK("gemm_f32_mfma_k");
    return VAE_OK;
}

static int linear_check(const Ctx& c, const vae_op& op) {
    const int K = op.d[0], N = op.d[1];
    if (K < 1 || N < 1 || op.d[2] < K || op.d[3] < N) VAE_FAIL(VAE_EINVAL, "linear: bad K=%d N=%d ld=%d,%d", K, N, op.d[2], op.d[3]);
    if (op.act != VAE_ACT_NONE && op.act != VAE_ACT_LEAKY) VAE_FAIL(VAE_EINVAL, "linear: unsupported activation");
    (void)c;
    return VAE_OK;
}
static inline int64_t strided_extent(int B, int ld, int n) { return (int64_t)(B - 1) * ld + n; }

static int linear_forward(const Ctx& c, const vae_op& op) {
    int rc = linear_check(c, op);
    if (rc) return rc;
    const int K = op.d[0], N = op.d[1], ldx = op.d[2], ldy = op.d[3];
    bool ok = true;
    GemmArgs a;
    a.A = resolve(c, op.r[0], strided_extent(c.B, ldx, K), "linear.in", &ok);
    a.C = resolve(c, op.r[1], strided_extent(c.B, ldy, N), "linear.out", &ok);
    a.Bm = resolve(c, op.r[4], (int64_t)N * K, "linear.w", &ok);
    a.bias = resolve_opt(c, op.r[5], N, "linear.b", &ok);
    if (!ok) return VAE_EINVAL;
    a.M = c.B; a.N = N; a.K = K;
    a.sam = ldx; a.sak = 1; a.sbk = 1; a.sbn = K; a.scm = ldy;
    a.act = op.act; a.slope = op.f[0]; a.mode = GEMM_STORE;
    return launch_gemm(c, a, 1);
}

static int linear_backward(const Ctx& c, const vae_op& op) {
    int rc = linear_check(c, op);
    if (rc) return rc;
    const int K = op.d[0], N = op.d[1], ldx = op.d[2], ldy = op.d[3], lddx = op.d[4], lddy = op.d[5];
    if (lddy < N || (op.r[2] >= 0 && lddx < K)) VAE_FAIL(VAE_EINVAL, "linear: bad grad leading dims");
    bool ok = true;
    const float* x = resolve(c, op.r[0], strided_extent(c.B, ldx, K), "linear.in", &ok);
    const float* y = resolve(c, op.r[1], strided_extent(c.B, ldy, N), "linear.out", &ok);
    float* dx = resolve_opt(c, op.r[2], strided_extent(c.B, lddx, K), "linear.din", &ok);
    float* dy = resolve(c, op.r[3], strided_extent(c.B, lddy, N), "linear.dout", &ok);
    const float* w = resolve(c, op.r[4], (int64_t)N * K, "linear.w", &ok);
    float* dw = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[4]), (int64_t)N * K, "linear.dw", &ok);
    float* db = op.r[5] < 0 ? nullptr : resolve(c, ((int64_t)2 << 56) | ref_off(op.r[5]), N, "linear.db", &ok);
    float* part = resolve(c, op.r[6], (int64_t)2 * N * VAE_MAX_CHUNKS, "linear.partials", &ok);
    if (!ok) return VAE_EINVAL;

    // 1) activation backward in place + bias gradient (column sums)
    if (op.act != VAE_ACT_NONE || db != nullptr) {
        int bchunk, nchunks;
        chunking(c.B, 64, &bchunk, &nchunks);
        hipLaunchKernelGGL(col_actbwd_reduce_k, dim3((N + 63) / 64, nchunks), dim3(256), 0, c.s, dy, y, part, c.B, N,
                           lddy, ldy, bchunk, op.act, op.f[0]);
        VAE_LAUNCH_CHECK("col_actbwd_reduce_k");
        if (db != nullptr) {
            hipLaunchKernelGGL(bias_grad_finalize_k, dim3((N + 63) / 64), dim3(64), 0, c.s, part, db, N, nchunks);
            VAE_LAUNCH_CHECK("bias_grad_finalize_k");
        }
    }
    // 2) dW[N][K] += dY^T X   (reduction over the batch: split-K + atomics into the zeroed grad)
    GemmArgs a;
    a.A = dy; a.Bm = x; a.C = dw; a.bias = nullptr;
    a.M = N; a.N = K; a.K = c.B;
    a.sam = 1; a.sak = lddy; a.sbk = ldx; a.sbn = 1; a.scm = K;
    a.act = VAE_ACT_NONE; a.slope = 0.f; a.mode = GEMM_ATOMIC;
    const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
    int ksplit = (int)ceil_div64(512, tiles);
    const int max_split = (int)ceil_div64(c.B, 256);
    if (ksplit > max_split) ksplit = max_split;
    rc = launch_gemm(c, a, ksplit);
    if (rc) return rc;
    // 3) dX[B][K] (=|+=) dY W
    if (dx != nullptr) {
        GemmArgs b;
        b.A = dy; b.Bm = w; b.C = dx; b.bias = nullptr;
        b.M = c.B; b.N = K; b.K = N;
        b.sam = lddy; b.sak = 1; b.sbk = K; b.sbn = 1; b.scm = lddx;
        b.act = VAE_ACT_NONE; b.slope = 0.f; b.mode = (op.flags & VAE_F_ACC_DIN) ? GEMM_ACCUM : GEMM_STORE;
        rc = launch_gemm(c, b, 1);
    }
    return rc;
}

// ---- reparam / slice / elbo ---------------------------------------------------------------
static int reparam_forward(const Ctx& c, const vae_op& op) {
    const int64_t n = (int64_t)c.B * op.d[0];
    if (op.d[0] < 1) VAE_FAIL(VAE_EINVAL, "reparam: bad latent");
    bool ok = true;
    const float* mu = resolve(c, op.r[0], n, "reparam.mu", &ok);
    const float* lv = resolve(c, op.r[1], n, "reparam.logvar", &ok);
    const float* eps = resolve(c, op.r[2], n, "reparam.eps", &ok);
    float* z = resolve(c, op.r[3], n, "reparam.z", &ok);
    if (!ok) return VAE_EINVAL;
    hipLaunchKernelGGL(reparam_fwd_k, dim3(ew_grid(n)), dim3(256), 0, c.s, mu, lv, eps, z, n);
    VAE_LAUNCH_CHECK("reparam_fwd_k");
    return VAE_OK;
}
static int reparam_backward(const Ctx& c, const vae_op& op) {
    const int64_t n = (int64_t)c.B * op.d[0];
    bool ok = true;
    const float* mu = resolve(c, op.r[0], n, "reparam.mu", &ok);
    const float* lv = resolve(c, op.r[1], n, "reparam.logvar", &ok);
    const float* eps = resolve(c, op.r[2], n, "reparam.eps", &ok);
    const float* dz = resolve(c, op.r[4], n, "reparam.dz", &ok);
    float* dmu = resolve(c, op.r[5], n, "reparam.dmu", &ok);
    float* dlv = resolve(c, op.r[6], n, "reparam.dlogvar", &ok);
    const float *gmu = nullptr, *glv = nullptr, *gs = nullptr;
    if (op.flags & VAE_F_EXT_GRAD) {
        gmu = resolve(c, op.r[7], n, "reparam.g_mu", &ok);
        glv = resolve(c, op.r[8], n, "reparam.g_logvar", &ok);
    }
    if (op.flags & VAE_F_KL_GRAD) gs = resolve_opt(c, op.r[9], 1, "reparam.gscale", &ok);
    if (!ok) return VAE_EINVAL;
    hipLaunchKernelGGL(reparam_bwd_k, dim3(ew_grid(n)), dim3(256), 0, c.s, mu, lv, eps, dz, gmu, glv, gs,
                       (op.flags & VAE_F_KL_GRAD) ? 1 : 0, dmu, dlv, n);
    VAE_LAUNCH_CHECK("reparam_bwd_k");
    return VAE_OK;
}

static int slice0_forward(const Ctx& c, const vae_op& op) {
    const int C = op.d[0], L = op.d[1];
    if (C < 1 || L < 1) VAE_FAIL(VAE_EINVAL, "slice0: bad shape");
    const int64_t rows = (int64_t)c.B * C;
    bool ok = true;
    const float* in = resolve(c, op.r[0], rows * L, "slice0.in", &ok);
    float* out = resolve(c, op.r[1], rows, "slice0.out", &ok);
    if (!ok) return VAE_EINVAL;
    hipLaunchKernelGGL(slice0_fwd_k, dim3(ew_grid(rows)), dim3(256), 0, c.s, in, out, rows, L);
    VAE_LAUNCH_CHECK("slice0_fwd_k");
    return VAE_OK;
}
static int slice0_backward(const Ctx& c, const vae_op& op) {
    const int C = op.d[0], L = op.d[1];
    const int64_t rows = (int64_t)c.B * C;
    bool ok = true;
    float* din = resolve(c, op.r[2], rows * L, "slice0.din", &ok);
    const float* dout = resolve(c, op.r[3], rows, "slice0.dout", &ok);
    if (!ok) return VAE_EINVAL;
    hipLaunchKernelGGL(slice0_bwd_k, dim3(ew_grid(rows * L)), dim3(256), 0, c.s, dout, din, rows * L, L);
    VAE_LAUNCH_CHECK("slice0_bwd_k");
    return VAE_OK;
}

static int elbo_fwd_impl(hipStream_t s, const float* xr, const float* tg, const float* mu, const float* lv, float* out,
                         float* partials, int B, int nps, int64_t t_sb, int64_t t_se, int latent, int bce,
                         int accumulate) {
    hipLaunchKernelGGL(elbo_fwd_k, dim3(VAE_ELBO_BLOCKS), dim3(256), 0, s, xr, tg, mu, lv, partials, (int64_t)B, nps,
                       t_sb, t_se, latent, bce);
    VAE_LAUNCH_CHECK("elbo_fwd_k");
    hipLaunchKernelGGL(elbo_finalize_k, dim3(1), dim3(256), 0, s, partials, out, VAE_ELBO_BLOCKS, accumulate);
    VAE_LAUNCH_CHECK("elbo_finalize_k");
    return VAE_OK;
}
static inline int64_t target_extent(int B, int nps, int64_t sb, int64_t se) {
    return (int64_t)(B - 1) * sb + (int64_t)(nps - 1) * se + 1;
}
static int elbo_forward(const Ctx& c, const vae_op& op) {
    const int nps = op.d[0], latent = op.d[3];
    if (nps < 1 || latent < 1 || op.d[1] < 0 || op.d[2] < 0) VAE_FAIL(VAE_EINVAL, "elbo: bad shape");
    bool ok = true;
    const float* xr = resolve(c, op.r[0], (int64_t)c.B * nps, "elbo.x_rec", &ok);
    const float* tg = resolve(c, op.r[1], target_extent(c.B, nps, op.d[1], op.d[2]), "elbo.target", &ok);
    const float* mu = resolve(c, op.r[2], (int64_t)c.B * latent, "elbo.mu", &ok);
    const float* lv = resolve(c, op.r[3], (int64_t)c.B * latent, "elbo.logvar", &ok);
    float* out = resolve(c, op.r[4], 4, "elbo.out", &ok);
    float* part = resolve(c, op.r[6], 2 * VAE_ELBO_BLOCKS, "elbo.partials", &ok);
    if (!ok) return VAE_EINVAL;
    return elbo_fwd_impl(c.s, xr, tg, mu, lv, out, part, c.B, nps, op.d[1], op.d[2], latent,
                         (op.flags & VAE_F_BCE) ? 1 : 0, op.d[5]);
}
static int elbo_backward(const Ctx& c, const vae_op& op) {
    const int nps = op.d[0], latent = op.d[3];
    bool ok = true;
    const float* xr = resolve(c, op.r[0], (int64_t)c.B * nps, "elbo.x_rec", &ok);
    const float* tg = resolve(c, op.r[1], target_extent(c.B, nps, op.d[1], op.d[2]), "elbo.target", &ok);
    float* dxr = resolve(c, op.r[5], (int64_t)c.B * nps, "elbo.dx_rec", &ok);
    const float* gs = resolve_opt(c, op.r[7], 1, "elbo.gscale", &ok);
    if (!ok) return VAE_EINVAL;
    // KL gradients are folded into the REPARAM op's backward (VAE_F_KL_GRAD)
    hipLaunchKernelGGL(elbo_bwd_k, dim3(ew_grid((int64_t)c.B * nps)), dim3(256), 0, c.s, xr, tg, nullptr, nullptr, gs,
                       dxr, nullptr, nullptr, (int64_t)c.B, nps, (int64_t)op.d[1], (int64_t)op.d[2], latent,
                       (op.flags & VAE_F_BCE) ? 1 : 0);
    VAE_LAUNCH_CHECK("elbo_bwd_k");
    return VAE_OK;
}

// ---- executor --------------------------------------------------------------------------------
static int make_ctx(Ctx* c, const vae_op* ops, int n_ops, float* const* bases, const int64_t* sizes, int n_bases,
                    int B, vae_stream_t stream) {
    if (ops == nullptr || n_ops < 0 || bases == nullptr || sizes == nullptr) VAE_FAIL(VAE_EINVAL, "null argument");
    if (n_bases < 4 || n_bases > 128) VAE_FAIL(VAE_EINVAL, "n_bases=%d out of range", n_bases);
    if (B < 1 || B > (1 << 24)) VAE_FAIL(VAE_EINVAL, "batch %d out of range", B);
    c->bases = bases; c->sizes = sizes; c->n_bases = n_bases; c->B = B; c->s = (hipStream_t)stream;
    return VAE_OK;
}

extern "C" int vae_run_forward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                               int n_bases, int B, int training, vae_stream_t stream) {
    Ctx c;
    int rc = make_ctx(&c, ops, n_ops, bases, base_sizes, n_bases, B, stream);
    for (int i = 0; i < n_ops && rc == VAE_OK; ++i) {
        const vae_op& op = ops[i];
        switch (op.kind) {
            case VAE_OP_CONV:
            case VAE_OP_CONVT: rc = conv_forward(c, op); break;
            case VAE_OP_BN: rc = bn_forward(c, op, training); break;
            case VAE_OP_LINEAR: rc = linear_forward(c, op); break;
            case VAE_OP_REPARAM: rc = reparam_forward(c, op); break;
            case VAE_OP_SLICE0: rc = slice0_forward(c, op); break;
            case VAE_OP_ELBO: rc = elbo_forward(c, op); break;
            default: vae_set_error("op %d: unknown kind %d", i, op.kind); rc = VAE_EINVAL;
        }
    }
    return rc;
}

extern "C" int vae_run_backward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                                int n_bases, int B, vae_stream_t stream) {
    Ctx c;
    int rc = make_ctx(&c, ops, n_ops, bases, base_sizes, n_bases, B, stream);
    if (rc) return rc;
    if (bases[2] == nullptr) VAE_FAIL(VAE_EINVAL, "backward needs the GRAD space (2) bound");
    // weight gradients accumulate with atomics: start from zero
    VAE_HIP(hipMemsetAsync(bases[2], 0, (size_t)base_sizes[2] * sizeof(float), c.s));
    for (int i = n_ops - 1; i >= 0 && rc == VAE_OK; --i) {
        const vae_op& op = ops[i];
        switch (op.kind) {
            case VAE_OP_CONV:
            case VAE_OP_CONVT: rc = conv_backward(c, op); break;
            case VAE_OP_BN: rc = bn_backward(c, op); break;
            case VAE_OP_LINEAR: rc = linear_backward(c, op); break;
            case VAE_OP_REPARAM: rc = reparam_backward(c, op); break;
            case VAE_OP_SLICE0: rc = slice0_backward(c, op); break;
            case VAE_OP_ELBO: rc = elbo_backward(c, op); break;
            default: vae_set_error("op %d: unknown kind %d", i, op.kind); rc = VAE_EINVAL;
        }
    }
    return rc;
}

// ---- standalone criterion, optimiser, rng ---------------------------------------------------
extern "C" int vae_elbo_forward(const float* x_rec, const float* target, const float* mu, const float* logvar,
                                float* out, float* partials, int B, int n_per_sample, int64_t t_sb, int64_t t_se,
                                int latent, int bce, vae_stream_t stream) {
    if (!x_rec || !target || !mu || !logvar || !out || !partials) VAE_FAIL(VAE_EINVAL, "elbo: null pointer");
    if (B < 1 || n_per_sample < 1 || latent < 1 || t_sb < 0 || t_se < 0) VAE_FAIL(VAE_EINVAL, "elbo: bad shape");
    return elbo_fwd_impl((hipStream_t)stream, x_rec, target, mu, logvar, out, partials, B, n_per_sample, t_sb, t_se,
                         latent, bce, 0);
}
extern "C" int vae_elbo_backward(const float* x_rec, const float* target, const float* mu, const float* logvar,
                                 const float* gscale, float* dx_rec, float* dmu, float* dlogvar, int B,
                                 int n_per_sample, int64_t t_sb, int64_t t_se, int latent, int bce,
                                 vae_stream_t stream) {
    if (!x_rec || !target || !mu || !logvar || !dx_rec || !dmu || !dlogvar) VAE_FAIL(VAE_EINVAL, "elbo: null pointer");
    if (B < 1 || n_per_sample < 1 || latent < 1 || t_sb < 0 || t_se < 0) VAE_FAIL(VAE_EINVAL, "elbo: bad shape");
    const int64_t n = (int64_t)B * (n_per_sample > latent ? n_per_sample : latent);
    hipLaunchKernelGGL(elbo_bwd_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x_rec, target, mu, logvar,
                       gscale, dx_rec, dmu, dlogvar, (int64_t)B, n_per_sample, t_sb, t_se, latent, bce);
    VAE_LAUNCH_CHECK("elbo_bwd_k");
    return VAE_OK;
}

extern "C" int vae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int32_t* tick, vae_stream_t stream) {
    if (!p || !g || !m || !v || !tick || n < 1) VAE_FAIL(VAE_EINVAL, "adam: bad argument");
    float* bc = reinterpret_cast<float*>(tick) + 2;  // tick block: {int32 t, pad, float bc1, float bc2}
    hipLaunchKernelGGL(adam_tick_k, dim3(1), dim3(64), 0, (hipStream_t)stream, tick, bc, beta1, beta2);
    VAE_LAUNCH_CHECK("adam_tick_k");
    hipLaunchKernelGGL(adam_step_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, bc);
    VAE_LAUNCH_CHECK("adam_step_k");
    return VAE_OK;
}

extern "C" int vae_randn(float* out, int64_t n, uint64_t* rng, vae_stream_t stream) {
    if (!out || !rng || n < 1) VAE_FAIL(VAE_EINVAL, "randn: bad argument");
    const int64_t nquad = (n + 3) / 4;
    hipLaunchKernelGGL(randn_k, dim3(ew_grid(nquad)), dim3(256), 0, (hipStream_t)stream, out, n, rng);
    VAE_LAUNCH_CHECK("randn_k");
    hipLaunchKernelGGL(rng_advance_k, dim3(1), dim3(64), 0, (hipStream_t)stream, rng, (uint64_t)nquad);
    VAE_LAUNCH_CHECK("rng_advance_k");
    return VAE_OK;
}

// ---- hipGraph capture -------------------------------------------------------------------------
static thread_local hipStream_t g_capture_stream = nullptr;

extern "C" int vae_graph_begin(vae_stream_t* capture_stream) {
    if (capture_stream == nullptr) VAE_FAIL(VAE_EINVAL, "graph_begin: null argument");
    if (g_capture_stream == nullptr) VAE_HIP(hipStreamCreateWithFlags(&g_capture_stream, hipStreamNonBlocking));
    VAE_HIP(hipStreamBeginCapture(g_capture_stream, hipStreamCaptureModeThreadLocal));
    *capture_stream = (vae_stream_t)g_capture_stream;
    return VAE_OK;
}
extern "C" int vae_graph_end(void** graph_handle) {
    if (graph_handle == nullptr || g_capture_stream == nullptr) VAE_FAIL(VAE_EINVAL, "graph_end: no capture in progress");
    hipGraph_t graph = nullptr;
    VAE_HIP(hipStreamEndCapture(g_capture_stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) VAE_FAIL(VAE_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *graph_handle = (void*)exec;
    return VAE_OK;
}
extern "C" int vae_graph_launch(void* graph_handle, vae_stream_t stream) {
    if (graph_handle == nullptr) VAE_FAIL(VAE_EINVAL, "graph_launch: null handle");
    VAE_HIP(hipGraphLaunch((hipGraphExec_t)graph_handle, (hipStream_t)stream));
    return VAE_OK;
}
extern "C" int vae_graph_destroy(void* graph_handle) {
    if (graph_handle != nullptr) VAE_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_handle));
    return VAE_OK;
}
