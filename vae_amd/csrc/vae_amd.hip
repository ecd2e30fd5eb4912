// vae_amd.hip - program executor and C ABI of libvae_amd.so (see include/vae_amd.h).
//
// Host code here only validates shapes/extents and enqueues kernels on the caller's stream:
// no allocation, no synchronisation, so any call sequence can be captured into a hipGraph
// (vae_graph_*), which is how the Python host removes per-kernel launch overhead from the
// train step.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <dlfcn.h>

#include <algorithm>
#include <mutex>

#include "chan.h"
#include "common.h"
#include "conv.h"
#include "elem.h"
#include "igemm.h"
#include "lin.h"
#include "wide.h"
#include "thin.h"
#include "scat.h"
#include "fc.h"
#include <stdlib.h>

static thread_local char g_err[512] = "";
void vae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vae_last_error(void) { return g_err; }

static int scratch_for_current_device(float** out);
extern "C" int vae_adam_step_hyper(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                   int32_t* tick, vae_stream_t stream);

extern "C" int vae_check_device(void) {
    int dev = 0;
    VAE_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    VAE_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        VAE_FAIL(VAE_ENODEV, "device %d is %s, libvae_amd is built for gfx950 only", dev, prop.gcnArchName);
    // the library's per-device set-up: resolve the device address of the static scratch (vae_elbo_forward_noalloc)
    float* p = nullptr;
    return scratch_for_current_device(&p);
}

// ---- reference resolution ------------------------------------------------------------------
// Data-parallel hook (vae_set_collective): SyncBN all-gathers one slab row per rank through it.
static vae_allgather_fn g_allgather = nullptr;
static void* g_allgather_user = nullptr;
static int g_rank = 0, g_world = 1;
static int g_bf16 = 0;  // vae_set_compute_dtype: 1 = bf16 MFMA operands (fp32 accumulate, fp32 storage)

struct Ctx {
    float* const* bases;
    const int64_t* sizes;
    int n_bases;
    int B;
    hipStream_t s;
    int fc;                   // runs of LINEAR / REPARAM ops go through fc_chain_k
    mutable int last_blocks;  // blocks that wrote the per-channel slab consumed by the next BN finalize
    mutable LinWgMulti lin_wg;  // Linear weight gradients deferred to one launch at the end of backward
    mutable WgFinMulti wg_fin;  // conv weight-gradient partial reductions, likewise
    // a virtual BatchNorm whose finalize has been handed to the prologue of its first consumer kernel (igemm.h: BnFin);
    // 0: none, 1: forward, 2: backward.  Whatever runs next and cannot take it launches bnv_finalize_*_k (bn_flush_pending).
    mutable int pend;
    mutable BnFin pend_fin;
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
// batch chunking for the two-stage channel reductions (grid = C x chunks workgroups)
static inline int chunking(int B, int C, int64_t per_sample, int* bchunk, int* nchunks) {
    // ~8k elements per workgroup, at least ~1024 workgroups over the channels, at most VAE_MAX_CHUNKS chunks
    int64_t want = ceil_div64((int64_t)B * per_sample, 8192);
    const int64_t fill = ceil_div64(1024, C);
    if (want < fill) want = fill;
    if (want > VAE_MAX_CHUNKS) want = VAE_MAX_CHUNKS;
    if (want < 1) want = 1;
    if (want > B) want = B;
    *bchunk = (int)ceil_div64(B, want);
    *nchunks = (int)ceil_div64(B, *bchunk);
    if ((int64_t)*bchunk * per_sample >= ((int64_t)1 << 31))  // chan_reduce_k indexes a chunk with 32 bits
        VAE_FAIL(VAE_EINVAL, "channel reduction: %lld elements per chunk exceed 2^31", (long long)*bchunk * per_sample);
    return VAE_OK;
}

// layers at least this wide (output rows x reduction length) take the K-loop MFMA kernels of wide.h
#define WIDE_MIN_N 16
#define WIDE_MIN_K 32
#define WIDE_WG_BLOCKS 1536  // workgroups the weight gradient's reduction is split into (6 per CU)
static FastDiv mk_fdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.mul = d > 1 ? (uint32_t)((((uint64_t)1 << 32) + d - 1) / d) : 0;
    return f;
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

// K = longest reduction of the launch (Cin x taps; per parity class for the transposed gather)
static bool wide_gather_ok(const ConvGeom& g, int64_t K, int64_t npix) {
    return g.Cout >= WIDE_MIN_N && K >= WIDE_MIN_K && K <= WD_KTAB_MAX && ceil_div64(npix, 64) < (1ll << 31) &&
           g.H < 16384 && g.W < 16384 && g.OH < 16384 && g.OW < 16384 && g.KH < 256 && g.KW < 256 &&
           (int64_t)g.Cin * g.H * g.W < (1ll << 30);
}
// y = conv(x): geometry in "conv orientation" (Cin,H,W) -> (Cout,OH,OW)
static int launch_conv_gather(const Ctx& c, const float* x, const float* w, const float* bias, float* y,
                              const ConvGeom& g, int act, float slope, int accumulate) {
    const int64_t npix = (int64_t)c.B * g.OH * g.OW;
    const int gx = (int)ceil_div64(npix, 256);
    const int64_t K = (int64_t)g.Cin * g.KH * g.KW;
    if (wide_gather_ok(g, K, npix)) {
        const int kpad = (int)ceil_div64(K, 16) * 16;
        if (g.Cout <= 32) {
            dim3 grid((unsigned)ceil_div64(npix, 64), (g.Cout + 31) / 32);
            if (g_bf16) hipLaunchKernelGGL((wide_gather_k<false, 1, true>), grid, dim3(WD_THREADS), 2 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
            else hipLaunchKernelGGL((wide_gather_k<false, 1>), grid, dim3(WD_THREADS), 2 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
        } else {
            dim3 grid((unsigned)ceil_div64(npix, 64), (g.Cout + 63) / 64);
            if (g_bf16) hipLaunchKernelGGL((wide_gather_k<false, 2, true>), grid, dim3(WD_THREADS), 2 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
            else hipLaunchKernelGGL((wide_gather_k<false, 2>), grid, dim3(WD_THREADS), 2 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
        }
        VAE_TNAME("wide_gather_k<false, %d, %s>", g.Cout <= 32 ? 1 : 2, g_bf16 ? "true" : "false");
        VAE_TFLOPS(2.0 * (double)npix * (double)K * g.Cout);
        VAE_TBYTES(4.0 * ((double)c.B * ((double)g.Cin * g.H * g.W + (double)g.Cout * g.OH * g.OW) + (double)K * g.Cout));
        VAE_LAUNCH_CHECK("wide_gather_k");
        return VAE_OK;
    }
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
    const int64_t K = (int64_t)g.Cin * ((g.KH + g.sh - 1) / g.sh) * ((g.KW + g.sw - 1) / g.sw);  // largest class
    if (wide_gather_ok(g, K, npix)) {
        const int kpad = (int)ceil_div64(K, 16) * 16;
        if (g.Cout <= 32) {
            dim3 grid((unsigned)ceil_div64(npix, 64), (g.Cout + 31) / 32, g.sh * g.sw);
            if (g_bf16) hipLaunchKernelGGL((wide_gather_k<true, 1, true>), grid, dim3(WD_THREADS), 3 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
            else hipLaunchKernelGGL((wide_gather_k<true, 1>), grid, dim3(WD_THREADS), 3 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
        } else {
            dim3 grid((unsigned)ceil_div64(npix, 64), (g.Cout + 63) / 64, g.sh * g.sw);
            if (g_bf16) hipLaunchKernelGGL((wide_gather_k<true, 2, true>), grid, dim3(WD_THREADS), 3 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
            else hipLaunchKernelGGL((wide_gather_k<true, 2>), grid, dim3(WD_THREADS), 3 * kpad * sizeof(int), c.s, x, w, bias, y,
                               c.B, g, act, slope, accumulate, kpad);
        }
        VAE_TNAME("wide_gather_k<true, %d, %s>", g.Cout <= 32 ? 1 : 2, g_bf16 ? "true" : "false");
        VAE_TFLOPS(2.0 * (double)c.B * g.Cin * g.H * g.W * (double)g.Cout * g.KH * g.KW);  // every input pixel meets every tap
        VAE_TBYTES(4.0 * ((double)c.B * ((double)g.Cin * g.H * g.W + (double)g.Cout * g.OH * g.OW) + (double)g.Cin * g.Cout * g.KH * g.KW));
        VAE_LAUNCH_CHECK("wide_gather_k");
        return VAE_OK;
    }
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
// split of the wide weight-gradient GEMM (wide.h): grid = chunks x ceil(A/64) x ceil(C*KH*KW/64)
struct WideWgPlan {
    bool ok;
    int gy, gz;
    int64_t chunks, chunk, R, NK;
};
static WideWgPlan wide_wgrad_plan(int B, int A, int SH, int SW, int C, int KH, int KW) {
    WideWgPlan p;
    p.R = (int64_t)B * SH * SW;
    p.NK = (int64_t)C * KH * KW;
    p.ok = A >= WIDE_MIN_N && p.NK >= WIDE_MIN_K && p.R < (1ll << 31) && p.R * SH * SW < (1ll << 32) && p.NK < (1 << 24);
    p.gy = (A + 63) / 64;
    p.gz = (int)ceil_div64(p.NK, 64);
    const int64_t want = ceil_div64(WIDE_WG_BLOCKS, (int64_t)p.gy * p.gz);
    p.chunk = ceil_div64(ceil_div64(p.R, want), 16) * 16;
    if (p.chunk < 256) p.chunk = 256;  // >= 16 K steps per workgroup
    p.chunks = ceil_div64(p.R, p.chunk);
    return p;
}
// dw[a][cc][kh][kw] += sum S[b,a,oh,ow] * G[b,cc,oh*sh-ph+kh,...]; part: the op's own partial region (r[10]) or nullptr
static int launch_wgrad(const Ctx& c, const float* S, const float* G, float* dw, int A, int SH, int SW, int C,
                        int GH, int GW, int KH, int KW, int sh, int sw, int ph, int pw, float* part) {
    const int64_t R = (int64_t)c.B * SH * SW;
    const WideWgPlan wp = wide_wgrad_plan(c.B, A, SH, SW, C, KH, KW);
    if (wp.ok) {
        WideWgArgs a;
        a.S = S; a.G = G; a.dw = dw; a.part = part;
        a.B = c.B; a.A = A; a.SH = SH; a.SW = SW; a.C = C; a.GH = GH; a.GW = GW; a.KH = KH; a.KW = KW;
        a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw;
        a.R = R; a.chunk = wp.chunk;
        a.d_SHW = mk_fdiv((uint32_t)(SH * SW)); a.d_SW = mk_fdiv((uint32_t)SW);
        a.d_KK = mk_fdiv((uint32_t)(KH * KW)); a.d_KW = mk_fdiv((uint32_t)KW);
        dim3 grid((unsigned)wp.chunks, wp.gy, wp.gz);
        if (g_bf16) hipLaunchKernelGGL(wide_wgrad_k<true>, grid, dim3(WD_THREADS), 0, c.s, a);
        else hipLaunchKernelGGL(wide_wgrad_k<false>, grid, dim3(WD_THREADS), 0, c.s, a);
        VAE_TNAME("wide_wgrad_k<%s>", g_bf16 ? "true" : "false");
        VAE_TFLOPS(2.0 * (double)R * A * (double)wp.NK);
        VAE_TBYTES(4.0 * ((double)R * A + (double)c.B * C * GH * GW));
        VAE_LAUNCH_CHECK("wide_wgrad_k");
        if (part != nullptr) {
            const int64_t n = (int64_t)A * wp.NK;
            hipLaunchKernelGGL(wide_wgrad_reduce_k, dim3((unsigned)ceil_div64(n, 64)), dim3(256), 0, c.s, part, dw, n,
                               (int)wp.chunks);
            VAE_LAUNCH_CHECK("wide_wgrad_reduce_k");
        }
        return VAE_OK;
    }
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
    if ((rc = chunking(c.B, g.Cout, (int64_t)g.OH * g.OW, &bchunk, &nchunks)) != VAE_OK) return rc;
    const float* x = resolve(c, op.r[0], nin, "conv.in", &ok);
    const float* y = resolve(c, op.r[1], nout, "conv.out", &ok);
    float* dx = resolve_opt(c, op.r[2], nin, "conv.din", &ok);
    float* dy = resolve(c, op.r[3], nout, "conv.dout", &ok);
    const float* w = resolve(c, op.r[4], nw, "conv.w", &ok);
    // gradients live at the same offsets in the GRAD space (space 2)
    float* dw = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[4]), nw, "conv.dw", &ok);
    float* db = op.r[5] < 0 ? nullptr : resolve(c, ((int64_t)2 << 56) | ref_off(op.r[5]), g.Cout, "conv.db", &ok);
    float* part = resolve(c, op.r[6], (int64_t)2 * g.Cout * VAE_MAX_CHUNKS, "conv.partials", &ok);
    // own region for the wide weight-gradient kernel's partial sums (without one: float atomics, not bit-reproducible;
    // program.py always provides it)
    const bool T = op.kind == VAE_OP_CONVT;
    const WideWgPlan wp = T ? wide_wgrad_plan(c.B, g.Cin, g.H, g.W, g.Cout, g.KH, g.KW)
                            : wide_wgrad_plan(c.B, g.Cout, g.OH, g.OW, g.Cin, g.KH, g.KW);
    float* wpart = (wp.ok && op.r[10] >= 0)
                       ? resolve(c, op.r[10], wp.chunks * wp.NK * (T ? g.Cin : g.Cout), "conv.wgrad_partials", &ok)
                       : nullptr;
    if (!ok) return VAE_EINVAL;

    // 1) dy <- dy * act'(y) in place, bias gradient = per-channel sum
    if (op.act != VAE_ACT_NONE || db != nullptr) {
        hipLaunchKernelGGL(chan_reduce_k<CR_ACTBWD>, dim3(g.Cout, nchunks), dim3(256), 0, c.s, nullptr, dy, y, nullptr,
                           part, c.B, g.Cout, g.OH * g.OW, bchunk, op.act, op.f[0], 0);
        VAE_TNAME("chan_reduce_k<2>");
        VAE_TBYTES(4.0 * 3.0 * (double)nout);
        VAE_LAUNCH_CHECK("chan_reduce_k<ACTBWD>");
        if (db != nullptr) {
            hipLaunchKernelGGL(bias_grad_finalize_k, dim3(g.Cout), dim3(64), 0, c.s, part, db, g.Cout,
                               nchunks);
            VAE_LAUNCH_CHECK("bias_grad_finalize_k");
        }
    }
    // 2) weight gradient, 3) input gradient
    if (op.kind == VAE_OP_CONV) {
        rc = launch_wgrad(c, dy, x, dw, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, wpart);
        if (rc) return rc;
        if (dx != nullptr) {
            ConvGeom t = g;  // transposed-conv orientation: (Cout,OH,OW) -> (Cin,H,W)
            t.Cin = g.Cout; t.H = g.OH; t.W = g.OW; t.Cout = g.Cin; t.OH = g.H; t.OW = g.W;
            rc = launch_convT_gather(c, dy, w, nullptr, dx, t, VAE_ACT_NONE, 0.f, op.flags & VAE_F_ACC_DIN);
        }
    } else {
        rc = launch_wgrad(c, x, dy, dw, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, wpart);
        if (rc) return rc;
        if (dx != nullptr) {
            ConvGeom t = g;  // conv orientation: (Cout,OH,OW) -> (Cin,H,W)
            t.Cin = g.Cout; t.H = g.OH; t.W = g.OW; t.Cout = g.Cin; t.OH = g.H; t.OW = g.W;
            rc = launch_conv_gather(c, dy, w, nullptr, dx, t, VAE_ACT_NONE, 0.f, op.flags & VAE_F_ACC_DIN);
        }
    }
    return rc;
}


// ---- fused implicit-GEMM conv path (igemm.h) ------------------------------------------------
#define IG_LDS_BUDGET_FLOATS 16000  // ~62.5 KiB dynamic LDS: two 512-thread workgroups per CU
#define IG_WG_LDS_BUDGET 16000
#define IG_MAX_GRID 512     // gather: persistent workgroups, two resident per CU, each loops over its chunks
#define IG_WG_MAX_GRID 512  // wgrad: same for kernels with <= WG_SMALL_T accumulator tiles, half of it for the fat ones
#define IG_SB_TARGET 512    // chunk = ceil(B / target) samples (bounded by the LDS budget and the prefetch registers):
                            // one chunk per workgroup at B = 4096 measured best on all three model families
#define IG_WG_SB_TARGET 512


// Geometry of a gather that reads (Cin,H,W) and produces (N,OH,OW).  transposed: ConvT forward /
// Conv dgrad; else Conv forward / ConvT dgrad.  wform_conv: weight element (ci,n) at
// n*Cin*KK + ci*KK (Conv layout seen from this gather), else at ci*N*KK + n*KK.
static IgGeom make_geom(int transposed, int Cin, int H, int W, int N, int OH, int OW, int KH, int KW, int sh, int sw,
                        int ph, int pw, int wform_conv) {
    IgGeom g;
    g.Cin = Cin; g.H = H; g.W = W; g.N = N; g.OH = OH; g.OW = OW; g.KH = KH; g.KW = KW;
    g.ph = ph; g.pw = pw; g.transposed = transposed;
    const int KK = KH * KW;
    if (!transposed) {
        g.si_h = sh; g.si_w = sw; g.so_h = 1; g.so_w = 1;
        g.halo_h = ph; g.halo_w = pw;
        g.Hp = (OH - 1) * sh + KH; if (g.Hp < H + ph) g.Hp = H + ph;
        g.Wp = (OW - 1) * sw + KW; if (g.Wp < W + pw) g.Wp = W + pw;
    } else {
        g.si_h = 1; g.si_w = 1; g.so_h = sh; g.so_w = sw;
        // enumerate classes: row offset dh = (rh + ph - kh)/sh over valid taps, rows a + dh, a < AH
        int lo_h = 0, hi_h = H - 1, lo_w = 0, hi_w = W - 1;
        for (int rh = 0; rh < sh; ++rh) {
            const int AH = (OH - rh + sh - 1) / sh;
            for (int kh = (rh + ph) % sh; kh < KH; kh += sh) {
                const int dh = (rh + ph - kh) / sh;
                if (dh < lo_h) lo_h = dh;
                if (AH - 1 + dh > hi_h) hi_h = AH - 1 + dh;
            }
        }
        for (int rw = 0; rw < sw; ++rw) {
            const int AW = (OW - rw + sw - 1) / sw;
            for (int kw = (rw + pw) % sw; kw < KW; kw += sw) {
                const int dw = (rw + pw - kw) / sw;
                if (dw < lo_w) lo_w = dw;
                if (AW - 1 + dw > hi_w) hi_w = AW - 1 + dw;
            }
        }
        g.halo_h = -lo_h; g.halo_w = -lo_w;
        g.Hp = hi_h - lo_h + 1; g.Wp = hi_w - lo_w + 1;
    }
    if (wform_conv) { g.w_cstride = KK; g.w_nstride = Cin * KK; }
    else { g.w_cstride = N * KK; g.w_nstride = KK; }
    g.Np = (N + 15) & ~15;
    g.d_HW = mk_fdiv((uint32_t)(H * W));
    g.d_W = mk_fdiv((uint32_t)W);
    g.d_CHW = mk_fdiv((uint32_t)(Cin * H * W));
    return g;
}

struct IgPlan {
    int SB, lds_in, lds_w, lds_out, lds_tab, lds_ptab, total, pre;
};
// Host mirror of the per-class tap enumeration (igemm.h IgClass); returns the class count.
static int fill_classes(const IgGeom& g, IgClass* cls, int* wsum, int* tsum) {
    const int ncls = g.so_h * g.so_w;
    int toff = 0, woff = 0;
    for (int i = 0; i < ncls; ++i) {
        IgClass& c = cls[i];
        c.rh = i / g.so_w; c.rw = i % g.so_w;
        c.AH = (g.OH - c.rh + g.so_h - 1) / g.so_h;
        c.AW = (g.OW - c.rw + g.so_w - 1) / g.so_w;
        if (c.AH < 0) c.AH = 0;
        if (c.AW < 0) c.AW = 0;
        if (g.transposed) {
            c.kh0 = (c.rh + g.ph) % g.so_h; c.kw0 = (c.rw + g.pw) % g.so_w;
            c.nkh = c.kh0 < g.KH ? (g.KH - c.kh0 + g.so_h - 1) / g.so_h : 0;
            c.nkw = c.kw0 < g.KW ? (g.KW - c.kw0 + g.so_w - 1) / g.so_w : 0;
        } else {
            c.kh0 = c.kw0 = 0; c.nkh = g.KH; c.nkw = g.KW;
        }
        c.ntaps = c.nkh * c.nkw;
        c.Kc = g.Cin * c.ntaps;
        c.KcPad = (c.Kc + 15) & ~15;  // the k-loop runs four MFMA steps per iteration
        c.toff = toff; c.woff = woff;
        c.d_ntaps = mk_fdiv(c.ntaps > 0 ? c.ntaps : 1);
        c.d_nkw = mk_fdiv(c.nkw > 0 ? c.nkw : 1);
        c.d_AHW = mk_fdiv(c.AH * c.AW > 0 ? c.AH * c.AW : 1);
        c.d_AW = mk_fdiv(c.AW > 0 ? c.AW : 1);
        toff += c.KcPad; woff += c.KcPad * g.Np;
    }
    *wsum = woff; *tsum = toff;
    return ncls;
}
// The kernel's time is that of the workgroup with the most chunks: with `rounds` = chunks per workgroup fixed by
// the largest feasible chunk, the smallest chunk that still needs only that many rounds wastes the least
// (B = 4096, 512 workgroups, 5 samples fit: 820 chunks = 2 rounds of 5; 4 samples = 1024 chunks = 2 rounds of 4).
static int balanced_chunk(int B, int sb, int grid) {
    const int64_t chunks = ceil_div64(B, sb), rounds = ceil_div64(chunks, grid);
    const int bal = (int)ceil_div64(B, (int64_t)grid * rounds);
    return bal < sb ? (bal < 1 ? 1 : bal) : sb;
}
// LDS need of the gather kernel for SB samples per chunk; picks SB for batch B (0 if it cannot fit)
static IgPlan plan_gather(const IgGeom& g, int B) {
    IgPlan p;
    memset(&p, 0, sizeof(p));
    IgClass cls[4];
    int wsum = 0, tsum = 0;
    if (g.so_h * g.so_w > 4) { p.SB = 0; return p; }
    fill_classes(g, cls, &wsum, &tsum);
    const int per_in = g.Cin * g.Hp * g.Wp, per_out = g.N * g.OH * g.OW + g.OH * g.OW;  // (+ pixel table)
    const int coef = 8 * (g.Cin + g.N) + g.Np + 8 + tsum + 4 * g.Np;  // (+ the set-up's weight-base table, one int per tap-table row; + the workgroup's running statistics)
    const int fixed = wsum + tsum + coef;
    int fit = (IG_LDS_BUDGET_FLOATS - fixed - 4) / (per_in + per_out);
    if (fit > 64) fit = 64;
    int sb = 0;
    if (fit >= 1) {
        sb = (B + IG_SB_TARGET - 1) / IG_SB_TARGET;  // about one chunk per resident workgroup (3-4 per CU): one latency round
        if (sb > fit) sb = fit;
        if (sb < 1) sb = 1;
        if (sb > B) sb = B;
        // register prefetch of the next chunk needs the chunk's input in IG_PRE registers per lane
        const int cap = IG_PRE * IG_THREADS / (g.Cin * g.H * g.W);
        p.pre = cap >= 1;
        if (p.pre && sb > cap) sb = cap;
        sb = balanced_chunk(B, sb, IG_MAX_GRID);
    }
    p.SB = sb;
    p.lds_in = (sb * per_in + 3) & ~3;  // keeps the weight image behind it 16-byte aligned
    p.lds_w = wsum; p.lds_out = sb * g.N * g.OH * g.OW; p.lds_tab = tsum;
    p.lds_ptab = sb * g.OH * g.OW;
    p.total = p.lds_in + p.lds_w + p.lds_out + p.lds_tab + p.lds_ptab + coef;
    if (p.lds_in >= 65536 || p.lds_out >= 65536) p.SB = 0;  // the pixel table packs two 16-bit LDS offsets
    // FastDiv exactness bound: n * d < 2^32 for every (n, d) pair the kernels divide
    const uint64_t nmax = (uint64_t)(p.lds_in > p.lds_out ? p.lds_in : p.lds_out) + wsum;
    if (nmax * (uint64_t)(g.Cin * g.H * g.W + g.N * g.OH * g.OW + 1) >= ((uint64_t)1 << 32)) p.SB = 0;
    return p;
}

struct WgPlan {
    int SB, lds_S, lds_G, lds_tab, total, Hp, Wp, MT, NT, pre;
};
// accumulator-tile buckets igemm_wgrad_k is instantiated for (MT x NT tiles of 16x16 per wave)
static int wgrad_bucket(int Sc, int NK, int* MT, int* NT) {
    static const int nts[] = {1, 2, 3, 5, 10};
    const int mt = (Sc + 15) / 16, ntt = (NK + 15) / 16;
    if (mt > 2) return 0;
    for (int nt : nts)
        if (nt >= ntt && !(mt == 2 && (nt == 1 || nt == 3))) { *MT = mt; *NT = nt; return 1; }
    return 0;
}
static WgPlan plan_wgrad(int B, int Sc, int SH, int SW, int Gc, int GH, int GW, int KH, int KW, int sh, int sw, int ph,
                         int pw) {
    WgPlan p;
    memset(&p, 0, sizeof(p));
    p.Hp = (SH - 1) * sh + KH; if (p.Hp < GH + ph) p.Hp = GH + ph;
    p.Wp = (SW - 1) * sw + KW; if (p.Wp < GW + pw) p.Wp = GW + pw;
    if (!wgrad_bucket(Sc, Gc * KH * KW, &p.MT, &p.NT) || Gc > 64) return p;
    const int T = p.MT * p.NT;
    const int per_S = Sc * SH * SW, per_G = Gc * p.Hp * p.Wp, per_tab = 2 * SH * SW;
    const int nred = 8 * (Sc + Gc) + IG_WAVES * 64;
    int fit = (IG_WG_LDS_BUDGET - nred - 16) / (per_S + per_G + per_tab);
    if (fit > 64) fit = 64;
    int sb = 0;
    if (fit >= 1) {
        sb = (B + IG_WG_SB_TARGET - 1) / IG_WG_SB_TARGET;
        while (sb < fit && sb * (per_S + per_G) < T * 256) ++sb;  // room for the cross-wave reduction image
        if (sb > fit) sb = fit;
        if (sb < 1) sb = 1;
        if (sb > B) sb = B;
        if (sb * (per_S + per_G) < T * 256 && (B >= fit)) sb = 0;
        // register prefetch of the next chunk: WG_PRE registers per lane and tensor
        const int big = per_S > Gc * GH * GW ? per_S : Gc * GH * GW;
        const int cap = WG_PRE_OF(T) * IG_THREADS / big;
        p.pre = cap >= 1 && sb >= 1;
        if (p.pre && sb > cap) sb = cap;
        if (sb >= 1) sb = balanced_chunk(B, sb, T <= WG_SMALL_T ? IG_WG_MAX_GRID : IG_WG_MAX_GRID / 2);
    }
    p.SB = sb;
    p.lds_S = sb * per_S; p.lds_G = sb * per_G;
    if (p.lds_S + p.lds_G < T * 256) p.lds_G = T * 256 - p.lds_S;  // tiny batches: pad the G region
    if ((p.lds_S + p.lds_G) & 1) ++p.lds_G;                        // the pixel table is int2: keep it 8-byte aligned
    p.lds_tab = sb * SH * SW;                                      // int2 entries
    p.total = p.lds_S + p.lds_G + 2 * p.lds_tab + nred + 8;
    if (p.total > IG_WG_LDS_BUDGET) p.SB = 0;
    return p;
}

template <int MT, int NT>
static void launch_wgrad_inst(const Ctx& c, const WgArgs& wa, int grid, size_t bytes) {
    if (g_bf16) hipLaunchKernelGGL((igemm_wgrad_k<MT, NT, true>), dim3(grid), dim3(IG_THREADS), bytes, c.s, wa);
    else hipLaunchKernelGGL((igemm_wgrad_k<MT, NT>), dim3(grid), dim3(IG_THREADS), bytes, c.s, wa);
}
// kernels with <= WG_SMALL_T accumulator tiles fit two workgroups per CU; the fat ones only one, and their
// per-workgroup reduction is expensive: fewer, longer-lived workgroups
static int wgrad_grid(const WgPlan& wp, int nchunks) {
    const int wg_max = wp.MT * wp.NT <= WG_SMALL_T ? IG_WG_MAX_GRID : IG_WG_MAX_GRID / 2;
    return nchunks < wg_max ? nchunks : wg_max;
}
static int64_t wgrad_part_floats(const WgPlan& wp, int B) {
    return (int64_t)wgrad_grid(wp, (B + wp.SB - 1) / wp.SB) * (wp.MT * wp.NT * 256 + 64);
}
static int flush_linear_wgrads(const Ctx& c);
static int flush_wgrad_finalize(const Ctx& c) {
    WgFinMulti& m = c.wg_fin;
    if (m.n == 0) {
        if (m.loss_part == nullptr) return VAE_OK;
        // a backward pass without a weight-gradient reduction: the loss sum gets a launch of its own
        hipLaunchKernelGGL(elbo_finalize_k, dim3(1), dim3(256), 0, c.s, m.loss_part, m.loss_out, m.loss_blocks, m.loss_accumulate);
        m.loss_part = nullptr;
        VAE_LAUNCH_CHECK("elbo_finalize_k");
        return VAE_OK;
    }
    int rc = flush_linear_wgrads(c);  // the Linear layers' partial rows must exist before they are reduced
    if (rc) return rc;
    int blocks = 0;
    for (int i = 0; i < m.n; ++i) {
        m.l[i].first_block = blocks;
        blocks += (m.l[i].ncols + 15) / 16;
    }
    // (+1: the workgroup that sums a pending elbo_fused_k's partial pairs)
    hipLaunchKernelGGL(wgrad_finalize_k, dim3(blocks + (m.loss_part != nullptr ? 1 : 0)), dim3(256), 0, c.s, m);
    m.loss_part = nullptr;
    if (vae_tl_timer != nullptr) {
        double fl = 0;
        for (int i = 0; i < m.n; ++i) fl += (double)(m.l[i].nrows + 2) * m.l[i].ncols;
        VAE_TBYTES(4.0 * fl);
    }
    m.n = 0;
    VAE_LAUNCH_CHECK("wgrad_finalize_k");
    return VAE_OK;
}

static int launch_wgrad(const Ctx& c, const WgArgs& wa, const WgPlan& wp, float* dw, float* db) {
    const int grid = wgrad_grid(wp, wa.nchunks);
    const size_t bytes = (size_t)wp.total * sizeof(float);
    switch (wp.MT * 100 + wp.NT) {
        case 101: launch_wgrad_inst<1, 1>(c, wa, grid, bytes); break;
        case 102: launch_wgrad_inst<1, 2>(c, wa, grid, bytes); break;
        case 103: launch_wgrad_inst<1, 3>(c, wa, grid, bytes); break;
        case 105: launch_wgrad_inst<1, 5>(c, wa, grid, bytes); break;
        case 110: launch_wgrad_inst<1, 10>(c, wa, grid, bytes); break;
        case 202: launch_wgrad_inst<2, 2>(c, wa, grid, bytes); break;
        case 205: launch_wgrad_inst<2, 5>(c, wa, grid, bytes); break;
        case 210: launch_wgrad_inst<2, 10>(c, wa, grid, bytes); break;
        default: VAE_FAIL(VAE_EINVAL, "conv wgrad: no kernel for %dx%d accumulator tiles", wp.MT, wp.NT);
    }
    VAE_TNAME("igemm_wgrad_k<%d, %d, %s>", wp.MT, wp.NT, g_bf16 ? "true" : "false");
    VAE_TBYTES(4.0 * (double)c.B * ((double)wa.Sc * wa.SH * wa.SW * (wa.S.mode >= TR_GRAD_BN ? 2 : 1) +
                                    (double)wa.Gc * wa.GH * wa.GW * (wa.G.mode >= TR_GRAD_BN ? 2 : 1)));
    VAE_LAUNCH_CHECK("igemm_wgrad_k");
    // the reduction of the partials is deferred: one launch for every conv layer of the pass (flush_wgrad_finalize)
    if (c.wg_fin.n == WG_FIN_MAX_LAYERS) {
        const int rc = flush_wgrad_finalize(c);
        if (rc) return rc;
    }
    WgFinLayer& L = c.wg_fin.l[c.wg_fin.n++];
    L.part = wa.part; L.dw = dw; L.dbias = db; L.nrows = grid; L.NT = wp.NT; L.T = wp.MT * wp.NT;
    L.Sc = wa.Sc; L.NK = wa.Gc * wa.KH * wa.KW; L.C = wa.bias_from_S ? wa.Sc : wa.Gc;
    L.plain = 0; L.stride = L.T * 256 + 64; L.ncols = L.stride;
    return VAE_OK;
}

// ---- streaming weight gradient of the mid-size layers (thin.h: mid_wgrad_k<1, 5>) ----
#define MID_MAX_GRID 1024
static int mid_grid(int B) {
    static int cap = 0;
    if (cap == 0) {
        const char* e = getenv("VAE_MID_GRID");
        cap = (e != nullptr && atoi(e) > 0) ? atoi(e) : 512;  // two workgroups per CU (the kernel holds ~210 registers)
        if (cap > MID_MAX_GRID) cap = MID_MAX_GRID;
    }
    return B < cap ? B : cap;
}
// S: Sc channels on SH x SW, G: Gc channels on GH x GW (roles as in igemm_wgrad_k).  Returns the samples per round
// (0: not eligible) and the instance: 1 = <1, 5, NS 4, NG 7> (8 <-> 16 channels), 2 = <2, 10, NS 5, NG 13> (16 <-> 32)
static int mid_wgrad_sbr(int Sc, int SH, int SW, int Gc, int GH, int GW, int KH, int KW, int MT, int NT, int* inst) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VAE_AMD_MID");
        on = (e != nullptr && *e) ? atoi(e) != 0 : 1;
    }
    *inst = 0;
    if (!on || Gc > 64 || Sc > 64) return 0;
    int ns, ng;
    if (MT == 1 && NT == 5) { *inst = 1; ns = 4; ng = 7; }
    else if (MT == 2 && NT == 10) { *inst = 2; ns = 5; ng = 13; }
    else return 0;
    const int shw = SH * SW, ns1 = Sc * shw, ng1 = Gc * GH * GW;
    if (shw < 1 || shw > 64) return 0;
    int sbr = 64 / shw;
    if (sbr * ns1 > 256 * ns) sbr = 256 * ns / ns1;
    if (sbr * ng1 > 256 * ng) sbr = 256 * ng / ng1;
    if (sbr < 1) *inst = 0;
    return sbr < 1 ? 0 : sbr;
}
static int64_t mid_part_floats(int B, int T) { return (int64_t)(B < MID_MAX_GRID ? B : MID_MAX_GRID) * (T * 256 + 64); }

// ---- thin outer layers (thin.h): first convolution 1 -> 8 and last transposed convolution 8 -> 1, 3x3 stride 2 pad 1 ----
#define TH_C 8
#define TH_MAX_GRID 2048
static int thin_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VAE_AMD_THIN");
        v = (e != nullptr && *e) ? atoi(e) != 0 : 1;
    }
    return v;
}
// 1: conv (Cin = 1 -> TH_C channels, output H x W = OH x OW); 2: transposed (TH_C channels H x W -> 1 channel 2H x 2W); 0: no
static int thin_kind(const ConvGeom& g, int T) {
    if (!thin_enabled() || g.KH != 3 || g.KW != 3 || g.sh != 2 || g.sw != 2 || g.ph != 1 || g.pw != 1) return 0;
    // (the single-channel image is fetched as one float4 per thread: rows a multiple of 4, at most 1024 elements)
    if (!T && g.Cin == 1 && g.Cout == TH_C && g.H == 2 * g.OH && g.W == 2 * g.OW && g.OH * g.OW <= TH_MAX_HW &&
        (g.W & 3) == 0 && g.H * g.W <= 4 * TH_THREADS)
        return 1;
    if (T && g.Cout == 1 && g.Cin == TH_C && g.OH == 2 * g.H && g.OW == 2 * g.W && g.H * g.W <= TH_MAX_HW &&
        (g.OW & 3) == 0 && g.OH * g.OW <= 4 * TH_THREADS)
        return 2;
    return 0;
}
static int thin_grid(int B) {
    static int cap = 0;
    if (cap == 0) {
        const char* e = getenv("VAE_THIN_GRID");
        cap = (e != nullptr && atoi(e) > 0) ? atoi(e) : 512;  // two workgroups per CU measured best (256: 0.672, 512: 0.652, 1024: 0.658 ms/step)
        if (cap > TH_MAX_GRID) cap = TH_MAX_GRID;
    }
    return B < cap ? B : cap;
}
static int64_t thin_part_floats(int B) { return (int64_t)(B < TH_MAX_GRID ? B : TH_MAX_GRID) * (10 * TH_C); }
static size_t thin_lds_bytes(int H, int W, int kernel) {  // kernel: 0 conv fwd, 1 conv wgrad, 2 convT fwd, 3 convT bwd
    const int img = (2 * H + 1) * (2 * W + 1);
    switch (kernel) {
        case 0: return (size_t)(2 * img) * sizeof(float);
        case 1: return (size_t)(2 * img + 8 * TH_C + 1024 + TH_THREADS * TH_MM_STRIDE + 16) * sizeof(float);
        case 2: return (size_t)(2 * TH_C * (H + 1) * (W + 1) + 8 * TH_C + BNFIN_SCR_FLOATS + 8) * sizeof(float);  // (+ finalize prologue scratch)
        default: return (size_t)(2 * img + 8 * TH_C + 1024 + TH_THREADS * TH_MM_STRIDE + 16) * sizeof(float);
    }
}
// the activation is a template argument of the thin kernels
#define THIN_LAUNCH(KERNEL, ACTV, GRID, BYTES, STREAM, ARGS)                                                                     \
    do {                                                                                                                       \
        if ((ACTV) == VAE_ACT_SIGMOID) hipLaunchKernelGGL((KERNEL<TH_C, VAE_ACT_SIGMOID>), dim3(GRID), dim3(TH_THREADS), BYTES, STREAM, ARGS); \
        else if ((ACTV) == VAE_ACT_LEAKY) hipLaunchKernelGGL((KERNEL<TH_C, VAE_ACT_LEAKY>), dim3(GRID), dim3(TH_THREADS), BYTES, STREAM, ARGS); \
        else hipLaunchKernelGGL((KERNEL<TH_C, VAE_ACT_NONE>), dim3(GRID), dim3(TH_THREADS), BYTES, STREAM, ARGS);             \
    } while (0)

static int conv_igemm_fits(const vae_op& op, int B) {
    const ConvGeom g = geom_of(op);
    const int T = op.kind == VAE_OP_CONVT;
    // forward gather, input-gradient gather, weight gradient
    const IgGeom f = make_geom(T, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, !T);
    const IgGeom d = make_geom(!T, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, T);
    const WgPlan w = T ? plan_wgrad(B, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw)
                       : plan_wgrad(B, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
    return plan_gather(f, B).SB >= 1 && plan_gather(d, B).SB >= 1 && w.SB >= 1 && g.sh <= 2 && g.sw <= 2;
}

extern "C" int vae_set_compute_dtype(int dtype) {
    if (dtype != VAE_DTYPE_F32 && dtype != VAE_DTYPE_BF16) VAE_FAIL(VAE_EINVAL, "set_compute_dtype: unknown dtype %d", dtype);
    g_bf16 = dtype == VAE_DTYPE_BF16;
    return VAE_OK;
}
extern "C" int vae_get_compute_dtype(void) { return g_bf16 ? VAE_DTYPE_BF16 : VAE_DTYPE_F32; }

extern "C" int vae_set_collective(vae_allgather_fn fn, void* user, int rank, int world) {
    if (fn == nullptr || world <= 1) {
        g_allgather = nullptr; g_allgather_user = nullptr; g_rank = 0; g_world = 1;
        return VAE_OK;
    }
    if (world > VAE_MAX_WORLD || rank < 0 || rank >= world) VAE_FAIL(VAE_EINVAL, "set_collective: bad rank %d / world %d", rank, world);
    g_allgather = fn; g_allgather_user = user; g_rank = rank; g_world = world;
    return VAE_OK;
}

// ---- data parallel: RCCL called from here -------------------------------------------------------
// The SyncBN all-gather and the gradient all-reduce enqueued on the caller's HIP stream by the library itself (no
// callback into the host language between the kernels of a step, `stream` honoured, so the step is one stream-ordered
// sequence that a hipGraph can capture).  librccl is opened at run time (the library has no link-time dependency on
// it; a single-GPU user never loads it); only the six entry points below are used, declared here with RCCL's ABI
// (rccl.h: ncclUniqueId is 128 opaque bytes passed by value, ncclFloat32 = 7, ncclSum = 0).
struct DpNcclId { char internal[128]; };
typedef int (*dp_get_unique_id_fn)(DpNcclId*);
typedef int (*dp_comm_init_rank_fn)(void**, int, DpNcclId, int);
typedef int (*dp_comm_destroy_fn)(void*);
typedef int (*dp_all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*dp_all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*dp_error_string_fn)(int);
static struct {
    void* handle;
    dp_get_unique_id_fn get_unique_id;
    dp_comm_init_rank_fn comm_init_rank;
    dp_comm_destroy_fn comm_destroy;
    dp_all_gather_fn all_gather;
    dp_all_reduce_fn all_reduce;
    dp_error_string_fn error_string;
    void* comm;       // SyncBN all-gathers (on the compute stream)
    void* comm_grad;  // gradient all-reduces (on the caller's communication stream, concurrent with the former)
    int rank, world;
    int single;       // 1: comm_grad == comm (vae_dp_set_single_comm)
} g_dp = {};
static std::mutex g_dp_mu;

static int dp_load(const char* path) {
    if (g_dp.handle != nullptr) return VAE_OK;
    void* h = nullptr;
    if (path != nullptr && path[0] != 0) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) VAE_FAIL(VAE_EINVAL, "data parallel: cannot open librccl (%s)", dlerror());
    g_dp.get_unique_id = (dp_get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
    g_dp.comm_init_rank = (dp_comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
    g_dp.comm_destroy = (dp_comm_destroy_fn)dlsym(h, "ncclCommDestroy");
    g_dp.all_gather = (dp_all_gather_fn)dlsym(h, "ncclAllGather");
    g_dp.all_reduce = (dp_all_reduce_fn)dlsym(h, "ncclAllReduce");
    g_dp.error_string = (dp_error_string_fn)dlsym(h, "ncclGetErrorString");
    if (!g_dp.get_unique_id || !g_dp.comm_init_rank || !g_dp.comm_destroy || !g_dp.all_gather || !g_dp.all_reduce ||
        !g_dp.error_string) {
        dlclose(h);
        VAE_FAIL(VAE_EINVAL, "data parallel: librccl lacks an entry point");
    }
    g_dp.handle = h;
    return VAE_OK;
}
#define VAE_RCCL(expr)                                                                              \
    do {                                                                                            \
        const int r_ = (expr);                                                                      \
        if (r_ != 0) VAE_FAIL(VAE_EHIP, "%s failed: %s", #expr, g_dp.error_string(r_));             \
    } while (0)

extern "C" int vae_dp_unique_id(const char* rccl_path, void* id256) {
    void* const id128 = id256;
    if (id128 == nullptr) VAE_FAIL(VAE_EINVAL, "dp_unique_id: null argument");
    std::lock_guard<std::mutex> lock(g_dp_mu);
    int rc = dp_load(rccl_path);
    if (rc) return rc;
    DpNcclId id[2];  // one per communicator
    VAE_RCCL(g_dp.get_unique_id(&id[0]));
    VAE_RCCL(g_dp.get_unique_id(&id[1]));
    memcpy(id128, id, sizeof(id));
    return VAE_OK;
}

// the hook that vae_dp_enable installs: in-place all-gather of `world` rows (row `rank` is filled)
static int dp_allgather_hook(void*, float* buf, int64_t n, vae_stream_t stream) {
    if (g_dp.comm == nullptr) { vae_set_error("data parallel: no communicator"); return 1; }
    const int r = g_dp.all_gather(buf + (int64_t)g_dp.rank * n, buf, (size_t)n, 7, g_dp.comm, (hipStream_t)stream);
    if (r != 0) { vae_set_error("ncclAllGather failed: %s", g_dp.error_string(r)); return 1; }
    return 0;
}

extern "C" int vae_dp_init(const char* rccl_path, const void* id256, int rank, int world) {
    const void* const id128 = id256;
    if (id128 == nullptr || world < 1 || world > VAE_MAX_WORLD || rank < 0 || rank >= world)
        VAE_FAIL(VAE_EINVAL, "dp_init: bad rank %d / world %d", rank, world);
    std::lock_guard<std::mutex> lock(g_dp_mu);
    int rc = dp_load(rccl_path);
    if (rc) return rc;
    if (g_dp.comm != nullptr) VAE_FAIL(VAE_EINVAL, "dp_init: a communicator exists already (vae_dp_shutdown first)");
    DpNcclId id[2];
    memcpy(id, id128, sizeof(id));
    void *comm = nullptr, *comm_grad = nullptr;
    VAE_RCCL(g_dp.comm_init_rank(&comm, world, id[0], rank));  // on the current device; every rank calls it
    if (g_dp.single) {
        comm_grad = comm;  // one communicator: the caller keeps every collective on one stream (vae_dp_set_single_comm)
    } else {
        const int r = g_dp.comm_init_rank(&comm_grad, world, id[1], rank);
        if (r != 0) {
            (void)g_dp.comm_destroy(comm);
            VAE_FAIL(VAE_EHIP, "ncclCommInitRank (gradient communicator) failed: %s", g_dp.error_string(r));
        }
    }
    g_dp.comm = comm; g_dp.comm_grad = comm_grad; g_dp.rank = rank; g_dp.world = world;
    return VAE_OK;
}

extern "C" int vae_dp_set_single_comm(int on) {
    std::lock_guard<std::mutex> lock(g_dp_mu);
    if (g_dp.comm != nullptr) VAE_FAIL(VAE_EINVAL, "dp_set_single_comm: a communicator exists already (vae_dp_shutdown first)");
    g_dp.single = on ? 1 : 0;
    return VAE_OK;
}
extern "C" int vae_dp_single_comm(void) { return g_dp.single; }

extern "C" int vae_dp_world(void) { return g_dp.comm != nullptr ? g_dp.world : 0; }

extern "C" int vae_dp_enable(int on) {
    if (!on) return vae_set_collective(nullptr, nullptr, 0, 1);
    if (g_dp.comm == nullptr) VAE_FAIL(VAE_EINVAL, "dp_enable: no communicator (vae_dp_init first)");
    return vae_set_collective(dp_allgather_hook, nullptr, g_dp.rank, g_dp.world);
}

extern "C" int vae_dp_all_gather_rows(float* buf, int64_t floats_per_rank, vae_stream_t stream) {
    if (buf == nullptr || floats_per_rank < 1) VAE_FAIL(VAE_EINVAL, "dp_all_gather_rows: bad arguments");
    return dp_allgather_hook(nullptr, buf, floats_per_rank, stream) == 0 ? VAE_OK : VAE_EHIP;
}

extern "C" int vae_dp_all_reduce_sum(float* buf, int64_t n, vae_stream_t stream) {
    if (buf == nullptr || n < 1) VAE_FAIL(VAE_EINVAL, "dp_all_reduce_sum: bad arguments");
    if (g_dp.comm == nullptr) VAE_FAIL(VAE_EINVAL, "dp_all_reduce_sum: no communicator (vae_dp_init first)");
    VAE_RCCL(g_dp.all_reduce(buf, buf, (size_t)n, 7, 0, g_dp.comm_grad, (hipStream_t)stream));
    return VAE_OK;
}

extern "C" int vae_dp_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_dp_mu);
    if (g_allgather == dp_allgather_hook) (void)vae_set_collective(nullptr, nullptr, 0, 1);
    if (g_dp.comm != nullptr) {
        void *comm = g_dp.comm, *comm_grad = g_dp.comm_grad;
        g_dp.comm = g_dp.comm_grad = nullptr; g_dp.world = 0; g_dp.rank = 0;
        // both communicators are destroyed whatever the first call returns; the first error is reported afterwards
        const int r0 = g_dp.comm_destroy(comm);
        const int r1 = comm_grad != comm ? g_dp.comm_destroy(comm_grad) : 0;
        if (r0 != 0 || r1 != 0)
            VAE_FAIL(VAE_EHIP, "ncclCommDestroy failed: %s", g_dp.error_string(r0 != 0 ? r0 : r1));
    }
    return VAE_OK;
}

extern "C" int64_t vae_conv_scratch_floats(const vae_op* op, int B) {
    if (op == nullptr || (op->kind != VAE_OP_CONV && op->kind != VAE_OP_CONVT) || B < 1) return 0;
    if (check_geom(*op) != VAE_OK) return 0;
    const ConvGeom g = geom_of(*op);
    const int T = op->kind == VAE_OP_CONVT;
    const WgPlan w = T ? plan_wgrad(B, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw)
                       : plan_wgrad(B, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
    if (w.SB < 1) return 0;
    // BatchNorm slabs: <= one row per sample, plus one gathered row per data-parallel rank
    const int64_t slab = (int64_t)4 * (g.Cin > g.Cout ? g.Cin : g.Cout) * ((int64_t)B + VAE_MAX_WORLD);
    return slab;
}

extern "C" int64_t vae_conv_wgrad_partial_floats(const vae_op* op, int B) {
    if (op == nullptr || (op->kind != VAE_OP_CONV && op->kind != VAE_OP_CONVT) || B < 1) return 0;
    if (check_geom(*op) != VAE_OK) return 0;
    const ConvGeom g = geom_of(*op);
    const int T = op->kind == VAE_OP_CONVT;
    if (!conv_igemm_fits(*op, B)) {  // not on the fused path: the wide weight-gradient kernel's [chunk][dw] partials
        const WideWgPlan p = T ? wide_wgrad_plan(B, g.Cin, g.H, g.W, g.Cout, g.KH, g.KW)
                               : wide_wgrad_plan(B, g.Cout, g.OH, g.OW, g.Cin, g.KH, g.KW);
        return p.ok ? p.chunks * p.NK * (T ? g.Cin : g.Cout) : 0;
    }
    const WgPlan w = T ? plan_wgrad(B, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw)
                       : plan_wgrad(B, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
    if (w.SB < 1) return 0;
    int64_t need = wgrad_part_floats(w, B);
    int mid_inst = 0;
    const bool mid_ok = (T ? mid_wgrad_sbr(g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, w.MT, w.NT, &mid_inst)
                           : mid_wgrad_sbr(g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, w.MT, w.NT, &mid_inst)) > 0;
    const int64_t thin = thin_kind(g, T) != 0 ? thin_part_floats(B) : 0, mid = mid_ok ? mid_part_floats(B, w.MT * w.NT) : 0;
    if (thin > need) need = thin;
    return mid > need ? mid : need;
}

// plain gathers that stream whole samples on mid_conv_k (thin.h) instead of the chunked kernel
static bool mid_conv_geom_ok(const IgGeom& q) {
    return !g_bf16 && !q.transposed && q.so_h == 1 && q.so_w == 1 && q.N <= 16 && q.Cin * q.KH * q.KW <= 4 * MIDC_KS &&
           q.OH * q.OW <= 64 && q.Cin * q.H * q.W <= 256 * MIDC_NE;
}
// transposed 3x3 stride-2 gathers that run as GEMM + col2im on scat_convT_k (scat.h); returns the samples per round
// (0: not this kernel) and the instance: 1 = <= 8 produced channels from <= 16, 2 = <= 16 from <= 32
static int scat_geom_sbr(const IgGeom& q, int* inst) {
    if (g_bf16 || !q.transposed || q.so_h != 2 || q.so_w != 2 || q.KH != 3 || q.KW != 3) return 0;
    const int cls = (q.N <= 8 && q.Cin <= 16) ? 1 : (q.N <= 16 && q.Cin <= 32) ? 2 : 0;
    if (cls == 0) return 0;
    const int HW = q.H * q.W, OHW = q.OH * q.OW, CHW = q.Cin * HW;
    if ((CHW & 3) || HW > 64 || OHW > TH_THREADS) return 0;
    int sbr = 64 / HW;
    if (TH_THREADS / OHW < sbr) sbr = TH_THREADS / OHW;
    const int nj = cls == 1 ? 1 : 2;
    if (nj * 4 * TH_THREADS / CHW < sbr) sbr = nj * 4 * TH_THREADS / CHW;
    if (sbr < 1) return 0;
    while (sbr & (sbr - 1)) sbr &= sbr - 1;  // a power of two: batches usually are, and the rounds then deal out evenly
    {   // (diagnostic: VAE_SCAT_SBR caps the samples per round)
        static const int cap = [] { const char* e = getenv("VAE_SCAT_SBR"); return (e != nullptr && atoi(e) > 0) ? atoi(e) : 0; }();
        if (cap > 0 && sbr > cap) sbr = cap;
    }
    *inst = cls;
    return sbr;
}
static bool scat_enabled() {
    static const int on = [] { const char* e = getenv("VAE_AMD_SCAT"); return (e != nullptr && *e) ? atoi(e) != 0 : 1; }();
    return on != 0;
}
extern "C" int vae_conv_fused_ok(const vae_op* op, int B);
static int gather_direct_nd(const IgGeom& g, const IgClass* cls, int ncls);
extern "C" int vae_conv_forward_variant(const vae_op* op, int B) {
    if (!vae_conv_fused_ok(op, B)) return 0;
    const ConvGeom g = geom_of(*op);
    const int T = op->kind == VAE_OP_CONVT;
    const IgGeom f = make_geom(T, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, !T);
    if (op->r[8] < 0 && thin_kind(g, T) != 0 && (thin_kind(g, T) == 2 || op->r[7] < 0)) return 300;  // thin.h streaming kernels
    {
        static const int mid_on = [] { const char* e = getenv("VAE_AMD_MID"); return (e != nullptr && *e) ? atoi(e) != 0 : 1; }();
        if (mid_on && mid_conv_geom_ok(f)) return 400;  // mid_conv_k
        int inst = 0;
        if (mid_on && scat_enabled() && scat_geom_sbr(f, &inst) > 0) return 500;  // scat_convT_k
    }
    IgClass cls[4];
    int wsum, tsum;
    const int ncls = fill_classes(f, cls, &wsum, &tsum);
    return (f.Np > 16 ? 200 : 100) + (f.Np > 16 ? 0 : gather_direct_nd(f, cls, ncls));
}

extern "C" int vae_conv_fused_ok(const vae_op* op, int B) {
    if (op == nullptr || (op->kind != VAE_OP_CONV && op->kind != VAE_OP_CONVT) || B < 1) return 0;
    if (check_geom(*op) != VAE_OK) return 0;
    return conv_igemm_fits(*op, B);
}

// thin layers (few output channels, short K) run on the vector ALU inside the same kernel (igemm.h, ND):
// returns the ND template argument, 0 = MFMA tiles
static int gather_direct_nd(const IgGeom& g, const IgClass* cls, int ncls) {
    int kmax = 0;
    for (int i = 0; i < ncls; ++i) kmax = cls[i].Kc > kmax ? cls[i].Kc : kmax;
    // one output channel: a 16-channel MFMA tile is 15/16 padding at any K (0.25 vs 0.7 cycles per pixel-tap)
    return g.N == 1 ? 1 : (kmax <= 36 && g.N <= 8) ? 8 : 0;
}

static int bn_flush_pending(const Ctx& c);
static int bn_take_pending(const Ctx& c, int dir, const float* coef, BnFin* out);
static int bnfin_enabled();
static_assert(VAE_BN_ACC_FLOATS(1) == BNACC_FLOATS(1), "include/vae_amd.h and igemm.h disagree on the accumulator block");
// the BatchNorm of C channels whose save block `ref` is (op flagged VAE_F_BN_ACC) takes fixed-point accumulators
static bool bn_acc_ok(const vae_op& op, int C) {
    return (op.flags & VAE_F_BN_ACC) && g_world <= 1 && C <= BNFIN_MAXC && bnfin_enabled();
}
// Puts a BnFin that the launch cannot use back as a launch of its own.
static int bn_untake(const Ctx& c, BnFin* f, int dir) {
    if (f->C == 0) return VAE_OK;
    c.pend = dir;
    c.pend_fin = *f;
    f->C = 0; f->slab = nullptr;
    return bn_flush_pending(c);
}
static int launch_igemm_gather(const Ctx& c, IgArgs& a, const char* what) {
    const IgPlan p = plan_gather(a.g, c.B);
    if (p.SB < 1) VAE_FAIL(VAE_EINVAL, "%s: layer does not fit the fused conv kernel's LDS budget", what);
    a.B = c.B; a.SB = p.SB; a.pre = p.pre;
    a.lds_in = p.lds_in; a.lds_w = p.lds_w; a.lds_out = p.lds_out; a.lds_tab = p.lds_tab; a.lds_ptab = p.lds_ptab;
    int wsum, tsum;
    a.ncls = fill_classes(a.g, a.cls, &wsum, &tsum);
    a.nchunks = (c.B + p.SB - 1) / p.SB;
    for (int i = 0, poff = 0; i < 4; ++i) {
        if (i < a.ncls) {
            a.cls[i].poff = poff;
            poff += p.SB * a.cls[i].AH * a.cls[i].AW;
        } else {  // unused classes start beyond everything (the kernel finds a lane's class by comparing starts)
            memset(&a.cls[i], 0, sizeof(IgClass));
            a.cls[i].poff = a.cls[i].woff = a.cls[i].toff = 0x7fffffff;
        }
    }
    {   // waves per class in proportion to tiles x K: start with one each, hand the rest to the most loaded
        int64_t work[4];
        int nw[4], left = IG_WAVES - a.ncls;
        for (int i = 0; i < a.ncls; ++i) {
            work[i] = (int64_t)((p.SB * a.cls[i].AH * a.cls[i].AW + 15) / 16) * (a.cls[i].KcPad > 0 ? a.cls[i].KcPad : 1);
            nw[i] = 1;
        }
        for (; left > 0; --left) {
            int best = 0;
            for (int i = 1; i < a.ncls; ++i)
                if (work[i] * nw[best] > work[best] * nw[i]) best = i;
            ++nw[best];
        }
        for (int i = 0, w0 = 0; i < a.ncls; ++i) {
            a.cls[i].w0 = w0; a.cls[i].nw = nw[i];
            w0 += nw[i];
        }
    }
    a.d_N = mk_fdiv(a.g.N);
    a.d_ntile = mk_fdiv(a.g.Np / 16);
    {   // mid-size plain gathers stream whole samples instead (thin.h: mid_conv_k)
        static const int mid_on = [] { const char* e = getenv("VAE_AMD_MID"); return (e != nullptr && *e) ? atoi(e) != 0 : 1; }();
        const IgGeom& q = a.g;
        const bool gradin = a.in.mode >= TR_GRAD_BN;
        const int actv = gradin ? a.in.act : a.act;
        if (mid_on && mid_conv_geom_ok(q) && (!gradin || a.act == VAE_ACT_NONE) && (a.ep_mode == EP_PLAIN || gradin)) {
            {
                const int frc = bn_untake(c, &a.fin, 1);  // (mid_conv_k has no finalize prologue)
                if (frc) return frc;
            }
            int grid = mid_grid(c.B);
            if (grid > a.nchunks) grid = a.nchunks;  // (the callers sized the slab for nchunks rows)
            const size_t mbytes = (size_t)(2 * q.Cin * q.Hp * q.Wp + 4 * q.Cin + 4 + 4 * q.N + 4 + 136) * sizeof(float);
            const bool epbn = a.ep_mode == EP_BNGRAD;
#define MIDC_LAUNCH(GI, EP, AC) hipLaunchKernelGGL((mid_conv_k<GI, EP, AC>), dim3(grid), dim3(TH_THREADS), mbytes, c.s, a)
#define MIDC_ACT(GI, EP)                                                   \
    do {                                                                   \
        if (actv == VAE_ACT_SIGMOID) MIDC_LAUNCH(GI, EP, VAE_ACT_SIGMOID); \
        else if (actv == VAE_ACT_LEAKY) MIDC_LAUNCH(GI, EP, VAE_ACT_LEAKY); \
        else MIDC_LAUNCH(GI, EP, VAE_ACT_NONE);                            \
    } while (0)
            if (!gradin) MIDC_ACT(false, false);
            else if (epbn) MIDC_ACT(true, true);
            else MIDC_ACT(true, false);
#undef MIDC_ACT
#undef MIDC_LAUNCH
            VAE_TNAME("mid_conv_k<%s, %s, %d>", gradin ? "true" : "false", (gradin && epbn) ? "true" : "false", actv);
            VAE_TBYTES(4.0 * ((double)c.B * ((double)q.Cin * q.H * q.W * (gradin ? 2 : 1) +
                                             (double)q.N * q.OH * q.OW * ((gradin && epbn) ? 2 : 1)) + (double)q.Cin * q.KH * q.KW * q.N));
            VAE_LAUNCH_CHECK("mid_conv_k");
            if (a.slab != nullptr) c.last_blocks = grid;
            return VAE_OK;
        }
    }
    {   // transposed 3x3 stride-2 gathers of mid-size layers: GEMM over the input pixels + col2im (scat.h)
        static const int mid_on = [] { const char* e = getenv("VAE_AMD_MID"); return (e != nullptr && *e) ? atoi(e) != 0 : 1; }();
        const IgGeom& q = a.g;
        const bool gradin = a.in.mode >= TR_GRAD_BN;
        const int actv = gradin ? a.in.act : a.act;
        int inst = 0;
        const int sbr = (mid_on && scat_enabled()) ? scat_geom_sbr(q, &inst) : 0;
        if (sbr > 0 && (!gradin || (a.act == VAE_ACT_NONE && a.bias == nullptr)) && (a.ep_mode == EP_PLAIN || gradin) &&
            (a.fin.C == 0 || (!gradin && a.fin.C == q.Cin))) {
            const int rounds = (c.B + sbr - 1) / sbr;
            static const int cap = [] { const char* e = getenv("VAE_SCAT_GRID"); return (e != nullptr && atoi(e) > 0) ? atoi(e) : 512; }();
            int grid = rounds < cap ? rounds : cap;
            if (grid > MID_MAX_GRID) grid = MID_MAX_GRID;
            if (grid > a.nchunks) grid = a.nchunks;  // (the callers sized the slab for nchunks rows)
            a.SB = sbr;
            const int co = inst == 1 ? 8 : 16, ks = inst == 1 ? 4 : 8, mp = SC_MP(co);
            const size_t sbytes = (size_t)(2 * 4 * ks * SC_BP + (inst == 1 ? 2 : 1) * (SC_ZROW + 1) * mp + 4 * q.Cin + 4 + 5 * co + 4 + 10 * co) * sizeof(float);  // (two product tiles for the pipelined 8-channel instances)
            const int ep = !gradin ? 0 : (a.ep_mode == EP_BNGRAD ? 2 : 1);
#define SCAT_LAUNCH(CO, KS, NJ, IN, EP, AC) hipLaunchKernelGGL((scat_convT_k<CO, KS, NJ, IN, EP, AC>), dim3(grid), dim3(TH_THREADS), sbytes, c.s, a)
#define SCAT_ACT(CO, KS, NJ, IN, EP)                                                     \
    do {                                                                                 \
        if (actv == VAE_ACT_SIGMOID) SCAT_LAUNCH(CO, KS, NJ, IN, EP, VAE_ACT_SIGMOID);   \
        else if (actv == VAE_ACT_LEAKY) SCAT_LAUNCH(CO, KS, NJ, IN, EP, VAE_ACT_LEAKY);  \
        else SCAT_LAUNCH(CO, KS, NJ, IN, EP, VAE_ACT_NONE);                              \
    } while (0)
#define SCAT_EP(CO, KS, NJ)                        \
    do {                                           \
        if (ep == 0) SCAT_ACT(CO, KS, NJ, 0, 0);   \
        else if (ep == 1) SCAT_ACT(CO, KS, NJ, 1, 1); \
        else SCAT_ACT(CO, KS, NJ, 1, 2);           \
    } while (0)
            if (inst == 1) SCAT_EP(8, 4, 1);
            else SCAT_EP(16, 8, 2);
#undef SCAT_EP
#undef SCAT_ACT
#undef SCAT_LAUNCH
            VAE_TNAME("scat_convT_k<%d, %d, %d, %d, %d, %d>", co, ks, inst == 1 ? 1 : 2, gradin ? 1 : 0, ep, actv);
            VAE_TBYTES(4.0 * ((double)c.B * ((double)q.Cin * q.H * q.W * (gradin ? 2 : 1) +
                                             (double)q.N * q.OH * q.OW * (ep == 2 ? 2 : 1)) + (double)q.Cin * q.KH * q.KW * q.N));
            VAE_LAUNCH_CHECK("scat_convT_k");
            if (a.slab != nullptr) c.last_blocks = grid;
            return VAE_OK;
        }
    }
    const int g_max = IG_MAX_GRID;
    const int grid = a.nchunks < g_max ? a.nchunks : g_max;
    const size_t bytes = (size_t)p.total * sizeof(float);
    if (a.fin.C > 0 && p.lds_out < BNFIN_SCR_FLOATS) {  // (the finalize prologue's scratch overlays the output tile)
        const int frc = bn_untake(c, &a.fin, 1);
        if (frc) return frc;
    }
    const bool grad = a.in.mode >= TR_GRAD_BN;
    const int nd = gather_direct_nd(a.g, a.cls, a.ncls);
    // (the ND > 0 instances compute on the vector ALU in fp32 whatever the compute dtype)
#define IG_LAUNCH(NT, GR, ND)                                                                                                  \
    do {                                                                                                                       \
        if (g_bf16 && ND == 0) hipLaunchKernelGGL((igemm_gather_k<NT, 2, GR, 0, true>), dim3(grid), dim3(IG_THREADS), bytes, c.s, a); \
        else hipLaunchKernelGGL((igemm_gather_k<NT, 2, GR, ND>), dim3(grid), dim3(IG_THREADS), bytes, c.s, a);                 \
    } while (0)
    if (a.g.Np > 16) {
        if (grad) IG_LAUNCH(2, true, 0); else IG_LAUNCH(2, false, 0);
    } else if (nd == 1) {
        if (grad) IG_LAUNCH(1, true, 1); else IG_LAUNCH(1, false, 1);
    } else if (nd == 8) {
        if (grad) IG_LAUNCH(1, true, 8); else IG_LAUNCH(1, false, 8);
    } else {
        if (grad) IG_LAUNCH(1, true, 0); else IG_LAUNCH(1, false, 0);
    }
#undef IG_LAUNCH
    VAE_TNAME("igemm_gather_k<%d, 2, %s, %d, %s>", a.g.Np > 16 ? 2 : 1, grad ? "true" : "false", a.g.Np > 16 ? 0 : nd,
              (g_bf16 && (a.g.Np > 16 || nd == 0)) ? "true" : "false");
    VAE_TBYTES(4.0 * ((double)c.B * ((double)a.g.Cin * a.g.H * a.g.W * (grad ? 2 : 1) +
                                     (double)a.g.N * a.g.OH * a.g.OW * (a.ep_mode == EP_BNGRAD ? 2 : 1)) +
                      (double)a.g.Cin * a.g.KH * a.g.KW * a.g.N));
    VAE_LAUNCH_CHECK(what);
    if (a.slab != nullptr) c.last_blocks = grid;  // one row per workgroup
    return VAE_OK;
}

static TensorIn tensor_plain(const float* p) {
    TensorIn t;
    t.src = p; t.src2 = nullptr; t.coef = nullptr; t.mode = TR_PLAIN; t.relu = 0; t.act = VAE_ACT_NONE; t.slope = 0.f;
    return t;
}

// scratch must hold grid * 2 * C floats for the slab
static int slab_ok(const Ctx& c, int64_t ref, int blocks, int C) {
    bool ok = true;
    resolve(c, ref, (int64_t)blocks * 4 * C, "conv.slab", &ok);
    return ok ? VAE_OK : VAE_EINVAL;
}

static int conv_forward_fused(const Ctx& c, const vae_op& op, int training) {
    int rc = check_geom(op);
    if (rc) return rc;
    const ConvGeom g = geom_of(op);
    const int T = op.kind == VAE_OP_CONVT;
    bool ok = true;
    const int64_t nin = (int64_t)c.B * g.Cin * g.H * g.W, nout = (int64_t)c.B * g.Cout * g.OH * g.OW;
    const int64_t nw = (int64_t)g.Cin * g.Cout * g.KH * g.KW;
    IgArgs a;
    a.g = make_geom(T, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, !T);
    a.in = tensor_plain(resolve(c, op.r[0], nin, "conv.in", &ok));
    if (op.r[7] >= 0) {
        a.in.mode = TR_BN;
        a.in.coef = resolve(c, op.r[7], (int64_t)8 * g.Cin, "conv.in_bn", &ok);
        a.in.relu = (op.flags & VAE_F_IN_RELU) ? 1 : 0;
    }
    a.out = resolve(c, op.r[1], nout, "conv.out", &ok);
    a.w = resolve(c, op.r[4], nw, "conv.w", &ok);
    a.bias = resolve_opt(c, op.r[5], g.Cout, "conv.b", &ok);
    a.act = op.act; a.slope = op.f[0];
    a.ep_mode = EP_PLAIN; a.ep_y = nullptr; a.ep_coef = nullptr; a.ep_relu = 0;
    a.slab = nullptr;
    if (!ok) return VAE_EINVAL;
    // the producer BatchNorm's pending finalize goes into this layer's kernel (or is launched now)
    if ((rc = bn_take_pending(c, 1, a.in.coef, &a.fin)) != VAE_OK) return rc;
    const int thin = op.r[8] < 0 ? thin_kind(g, T) : 0;  // (the thin kernels take no BatchNorm statistics of their output)
    if (a.fin.C > 0 && thin == 1 && (rc = bn_untake(c, &a.fin, 1)) != VAE_OK) return rc;
    if (thin != 0 && (thin == 2 || op.r[7] < 0)) {
        ThinArgs t;
        memset(&t, 0, sizeof(t));
        t.fin = a.fin;
        t.in = a.in; t.w = a.w; t.bias = a.bias; t.out = a.out; t.B = c.B; t.act = op.act; t.slope = op.f[0];
        t.H = thin == 1 ? g.OH : g.H; t.W = thin == 1 ? g.OW : g.W;
        t.d_W = mk_fdiv((uint32_t)t.W); t.d_W4 = mk_fdiv((uint32_t)(2 * t.W / 4));
        if (thin == 1) THIN_LAUNCH(thin_conv_fwd_k, op.act, thin_grid(c.B), thin_lds_bytes(t.H, t.W, 0), c.s, t);
        else THIN_LAUNCH(thin_convT_fwd_k, op.act, thin_grid(c.B), thin_lds_bytes(t.H, t.W, 2), c.s, t);
        VAE_TNAME("%s<8, %d>", thin == 1 ? "thin_conv_fwd_k" : "thin_convT_fwd_k", op.act);
        VAE_TBYTES(4.0 * ((double)nin + (double)nout + (double)nw));
        VAE_LAUNCH_CHECK("thin conv forward");
        return VAE_OK;
    }
    a.acc_save = nullptr; a.acc_pivot = nullptr;
    if (op.r[8] >= 0 && training) {
        const IgPlan p = plan_gather(a.g, c.B);
        if (p.SB >= 1 && slab_ok(c, op.r[6], (c.B + p.SB - 1) / p.SB, g.Cout)) return VAE_EINVAL;
        a.slab = resolve(c, op.r[6], 4 * g.Cout, "conv.slab", &ok);
        if (bn_acc_ok(op, g.Cout) && op.r[9] >= 0) {  // the statistics also go to the BatchNorm's accumulators
            a.acc_save = resolve(c, op.r[8], (int64_t)8 * g.Cout + BNACC_FLOATS(g.Cout), "conv.out_bn+accumulators", &ok);
            a.acc_pivot = resolve(c, op.r[9], g.Cout, "conv.out_bn.running_mean", &ok);
        }
    }
    if (!ok) return VAE_EINVAL;
    return launch_igemm_gather(c, a, "igemm_gather_k(fwd)");
}

static int conv_backward_fused(const Ctx& c, const vae_op& op) {
    int rc = check_geom(op);
    if (rc) return rc;
    const ConvGeom g = geom_of(op);
    const int T = op.kind == VAE_OP_CONVT;
    bool ok = true;
    const int64_t nin = (int64_t)c.B * g.Cin * g.H * g.W, nout = (int64_t)c.B * g.Cout * g.OH * g.OW;
    const int64_t nw = (int64_t)g.Cin * g.Cout * g.KH * g.KW;
    // layer input as the forward saw it
    TensorIn tin = tensor_plain(resolve(c, op.r[0], nin, "conv.in", &ok));
    if (op.r[7] >= 0) {
        tin.mode = TR_BN;
        tin.coef = resolve(c, op.r[7], (int64_t)8 * g.Cin, "conv.in_bn", &ok);
        tin.relu = (op.flags & VAE_F_IN_RELU) ? 1 : 0;
    }
    // d(pre-activation) of this layer, computed on load from the stored gradient and the output
    TensorIn tg = tensor_plain(resolve(c, op.r[3], nout, "conv.dout", &ok));
    tg.src2 = resolve(c, op.r[1], nout, "conv.out", &ok);
    tg.act = op.act; tg.slope = op.f[0];
    if (op.r[8] >= 0) {
        tg.mode = TR_GRAD_BN;
        tg.coef = resolve(c, op.r[8], (int64_t)8 * g.Cout, "conv.out_bn", &ok);
    } else if (op.act != VAE_ACT_NONE) {
        tg.mode = TR_GRAD_ACT;
    }
    const float* w = resolve(c, op.r[4], nw, "conv.w", &ok);
    float* dw = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[4]), nw, "conv.dw", &ok);
    float* db = op.r[5] < 0 ? nullptr : resolve(c, ((int64_t)2 << 56) | ref_off(op.r[5]), g.Cout, "conv.db", &ok);
    float* dx = resolve_opt(c, op.r[2], nin, "conv.din", &ok);
    if (!ok) return VAE_EINVAL;

    if (op.r[8] < 0 && (rc = bn_flush_pending(c)) != VAE_OK) return rc;
    {   // thin outer layers: one streaming kernel for the whole backward of the layer (thin.h)
        const int thin = op.r[8] < 0 ? thin_kind(g, T) : 0;
        if ((thin == 1 && dx == nullptr && op.r[7] < 0) || thin == 2) {
            ThinArgs t;
            memset(&t, 0, sizeof(t));
            t.in = tin; t.g = tg; t.w = w; t.B = c.B; t.want_bias = db != nullptr;
            t.H = thin == 1 ? g.OH : g.H; t.W = thin == 1 ? g.OW : g.W;
            t.d_W = mk_fdiv((uint32_t)t.W); t.d_W4 = mk_fdiv((uint32_t)(2 * t.W / 4));
            t.slope = op.f[0];
            const int grid = thin_grid(c.B);
            const int rowlen = thin == 1 ? 10 * TH_C : 9 * TH_C + 1;
            t.part = resolve(c, op.r[10], (int64_t)grid * rowlen, "conv.wgrad_partials", &ok);
            if (!ok) return VAE_EINVAL;
            if (thin == 1) {
                THIN_LAUNCH(thin_conv_wgrad_k, op.act, grid, thin_lds_bytes(t.H, t.W, 1), c.s, t);
            } else {
                t.out = dx;
                if (dx != nullptr && op.r[7] >= 0) {  // the producer's BatchNorm is virtual: its backward sums, one row per workgroup
                    if (slab_ok(c, op.r[6], grid, g.Cin)) return VAE_EINVAL;
                    t.slab = resolve(c, op.r[6], 4 * g.Cin, "conv.slab", &ok);
                    t.ep_coef = tin.coef; t.ep_relu = tin.relu;
                    if (bn_acc_ok(op, g.Cin))
                        t.acc_save = resolve(c, op.r[7], (int64_t)8 * g.Cin + BNACC_FLOATS(g.Cin), "conv.in_bn+accumulators", &ok);
                    if (!ok) return VAE_EINVAL;
                }
                THIN_LAUNCH(thin_convT_bwd_k, op.act, grid, thin_lds_bytes(t.H, t.W, 3), c.s, t);
                if (t.slab != nullptr) c.last_blocks = grid;
            }
            VAE_TNAME("%s<8, %d>", thin == 1 ? "thin_conv_wgrad_k" : "thin_convT_bwd_k", op.act);
            // conv1: x, gradient and output read once; convT3: gradient, output and input read, input gradient written
            VAE_TBYTES(4.0 * (thin == 1 ? (double)nin + 2.0 * nout : 2.0 * nout + (double)nin * (dx != nullptr ? 2 : 1)));
            VAE_LAUNCH_CHECK("thin conv backward");
            if (c.wg_fin.n == WG_FIN_MAX_LAYERS && (rc = flush_wgrad_finalize(c)) != VAE_OK) return rc;
            WgFinLayer& F = c.wg_fin.l[c.wg_fin.n++];
            memset(&F, 0, sizeof(F));
            F.part = t.part; F.dw = dw; F.dbias = db; F.nrows = grid; F.NT = 1; F.T = 0;
            F.Sc = TH_C; F.NK = 9; F.C = thin == 1 ? TH_C : 1;
            F.plain = 1; F.stride = rowlen; F.ncols = 9 * TH_C + (db != nullptr ? F.C : 0);
            return VAE_OK;
        }
    }

    // 1) weight (+bias) gradient
    WgArgs wa;
    WgPlan wp;
    if (!T) {
        wp = plan_wgrad(c.B, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
        wa.S = tg; wa.G = tin; wa.bias_from_S = 1;
        wa.Sc = g.Cout; wa.SH = g.OH; wa.SW = g.OW; wa.Gc = g.Cin; wa.GH = g.H; wa.GW = g.W;
    } else {
        wp = plan_wgrad(c.B, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw);
        wa.S = tin; wa.G = tg; wa.bias_from_S = 0;
        wa.Sc = g.Cin; wa.SH = g.H; wa.SW = g.W; wa.Gc = g.Cout; wa.GH = g.OH; wa.GW = g.OW;
    }
    if (wp.SB < 1) VAE_FAIL(VAE_EINVAL, "conv wgrad: layer does not fit the fused kernel");
    wa.part = resolve(c, op.r[10], wgrad_part_floats(wp, c.B), "conv.wgrad_partials", &ok);  // the layer's own region
    if (!ok) return VAE_EINVAL;
    wa.pre = wp.pre;
    wa.want_bias = db != nullptr; wa.B = c.B; wa.SB = wp.SB;
    wa.KH = g.KH; wa.KW = g.KW; wa.sh = g.sh; wa.sw = g.sw; wa.ph = g.ph; wa.pw = g.pw;
    wa.Hp = wp.Hp; wa.Wp = wp.Wp;
    wa.dS_HW = mk_fdiv(wa.SH * wa.SW); wa.dS_W = mk_fdiv(wa.SW); wa.dS_CHW = mk_fdiv(wa.Sc * wa.SH * wa.SW);
    wa.dG_HW = mk_fdiv(wa.GH * wa.GW); wa.dG_W = mk_fdiv(wa.GW); wa.dG_CHW = mk_fdiv(wa.Gc * wa.GH * wa.GW);
    wa.d_KK = mk_fdiv(g.KH * g.KW); wa.d_KW = mk_fdiv(g.KW);
    wa.d_bC = mk_fdiv(wa.bias_from_S ? wa.Sc : wa.Gc);
    wa.lds_S = wp.lds_S; wa.lds_G = wp.lds_G; wa.lds_tab = wp.lds_tab;
    wa.nchunks = (c.B + wp.SB - 1) / wp.SB;
    int mid_inst = 0;
    const int mid_sbr = g_bf16 ? 0 : mid_wgrad_sbr(wa.Sc, wa.SH, wa.SW, wa.Gc, wa.GH, wa.GW, g.KH, g.KW, wp.MT, wp.NT, &mid_inst);
    if (!(mid_sbr > 0 && (op.act == VAE_ACT_NONE || op.act == VAE_ACT_LEAKY)) && (rc = bn_flush_pending(c)) != VAE_OK) return rc;
    if (mid_sbr > 0 && (op.act == VAE_ACT_NONE || op.act == VAE_ACT_LEAKY)) {
        MidWgArgs m;
        memset(&m, 0, sizeof(m));
        // the BatchNorm behind this layer: its backward finalize runs in this kernel's prologue
        if ((rc = bn_take_pending(c, 2, tg.coef, &m.fin)) != VAE_OK) return rc;
        m.S = wa.S; m.G = wa.G; m.want_bias = db != nullptr; m.B = c.B; m.SBR = mid_sbr;
        m.Sc = wa.Sc; m.SH = wa.SH; m.SW = wa.SW; m.Gc = wa.Gc; m.GH = wa.GH; m.GW = wa.GW;
        m.KH = g.KH; m.KW = g.KW; m.sh = g.sh; m.sw = g.sw; m.ph = g.ph; m.pw = g.pw; m.Hp = wp.Hp; m.Wp = wp.Wp;
        m.slope = op.f[0];
        m.dS_HW = mk_fdiv(wa.SH * wa.SW); m.dS_W = mk_fdiv(wa.SW); m.dS_CHW = mk_fdiv(wa.Sc * wa.SH * wa.SW);
        m.dG_HW = mk_fdiv(wa.GH * wa.GW); m.dG_W = mk_fdiv(wa.GW); m.dG_CHW = mk_fdiv(wa.Gc * wa.GH * wa.GW);
        m.d_KK = mk_fdiv(g.KH * g.KW); m.d_KW = mk_fdiv(g.KW);
        int grid = mid_grid((c.B + mid_sbr - 1) / mid_sbr);
        if (mid_inst == 2 && grid > 256) grid = 256;  // (20 accumulator tiles: one workgroup per CU; 512: 36 us, 256: 25 us)
        const int TE = wp.MT * wp.NT * 256;
        m.part = resolve(c, op.r[10], (int64_t)grid * (TE + 64), "conv.wgrad_partials", &ok);
        if (!ok) return VAE_EINVAL;
        const int nS = mid_sbr * wa.Sc * wa.SH * wa.SW, imgG = mid_sbr * wa.Gc * wp.Hp * wp.Wp;
        const int nGr = mid_sbr * wa.Gc * wa.GH * wa.GW;
        int fl = 2 * nS + 2 * imgG + 4 * (wa.Sc + wa.Gc) + 8 + BNFIN_SCR_FLOATS + 8;  // (+ the finalize prologue's scratch)
        if (fl < 4 * 5 * 256) fl = 4 * 5 * 256;          // the cross-wave reduction's passes of five tiles
        if (fl < (nS > nGr ? nS : nGr)) fl = nS > nGr ? nS : nGr;  // the bias partials of a round
        const size_t bytes = (size_t)fl * sizeof(float);
#define MIDW_LAUNCH(MTv, NTv, NSv, NGv)                                                                                              \
    do {                                                                                                                           \
        if (!T) {  /* Conv: S is the gradient */                                                                                   \
            if (op.act == VAE_ACT_LEAKY) hipLaunchKernelGGL((mid_wgrad_k<MTv, NTv, NSv, NGv, true, VAE_ACT_LEAKY>), dim3(grid), dim3(TH_THREADS), bytes, c.s, m); \
            else hipLaunchKernelGGL((mid_wgrad_k<MTv, NTv, NSv, NGv, true, VAE_ACT_NONE>), dim3(grid), dim3(TH_THREADS), bytes, c.s, m); \
        } else {                                                                                                                   \
            if (op.act == VAE_ACT_LEAKY) hipLaunchKernelGGL((mid_wgrad_k<MTv, NTv, NSv, NGv, false, VAE_ACT_LEAKY>), dim3(grid), dim3(TH_THREADS), bytes, c.s, m); \
            else hipLaunchKernelGGL((mid_wgrad_k<MTv, NTv, NSv, NGv, false, VAE_ACT_NONE>), dim3(grid), dim3(TH_THREADS), bytes, c.s, m); \
        }                                                                                                                          \
    } while (0)
        if (bytes > 64 * 1024) VAE_FAIL(VAE_EINVAL, "mid wgrad: %zu bytes of LDS", bytes);
        if (mid_inst == 1) MIDW_LAUNCH(1, 5, 4, 7);
        else MIDW_LAUNCH(2, 10, 5, 13);
#undef MIDW_LAUNCH
        VAE_TNAME("mid_wgrad_k<%d, %d, %d, %d, %s, %d>", wp.MT, wp.NT, mid_inst == 1 ? 4 : 5, mid_inst == 1 ? 7 : 13,
                  !T ? "true" : "false", op.act == VAE_ACT_LEAKY ? 1 : 0);
        VAE_TBYTES(4.0 * (double)c.B * ((double)g.Cin * g.H * g.W + 2.0 * g.Cout * g.OH * g.OW));
        VAE_LAUNCH_CHECK("mid_wgrad_k");
        if (c.wg_fin.n == WG_FIN_MAX_LAYERS && (rc = flush_wgrad_finalize(c)) != VAE_OK) return rc;
        WgFinLayer& L = c.wg_fin.l[c.wg_fin.n++];
        L.part = m.part; L.dw = dw; L.dbias = db; L.nrows = grid; L.NT = wp.NT; L.T = wp.MT * wp.NT;
        L.Sc = wa.Sc; L.NK = wa.Gc * g.KH * g.KW; L.C = wa.bias_from_S ? wa.Sc : wa.Gc;
        L.plain = 0; L.stride = L.T * 256 + 64; L.ncols = L.stride;
    } else {
        rc = launch_wgrad(c, wa, wp, dw, db);
        if (rc) return rc;
    }

    // 2) input gradient
    if (dx == nullptr) return VAE_OK;
    IgArgs a;
    a.g = make_geom(!T, g.Cout, g.OH, g.OW, g.Cin, g.H, g.W, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, T);
    a.in = tg;
    a.out = dx; a.w = w; a.bias = nullptr; a.act = VAE_ACT_NONE; a.slope = 0.f;
    a.ep_mode = EP_PLAIN; a.ep_y = nullptr; a.ep_coef = nullptr; a.ep_relu = 0; a.slab = nullptr;
    memset(&a.fin, 0, sizeof(a.fin));
    a.acc_save = nullptr; a.acc_pivot = nullptr;
    if (op.r[7] >= 0) {  // the producer's BatchNorm is virtual: mask by its ReLU and take its backward sums here
        a.ep_mode = EP_BNGRAD;
        a.ep_y = tin.src; a.ep_coef = tin.coef; a.ep_relu = tin.relu;
        const IgPlan p = plan_gather(a.g, c.B);
        if (p.SB >= 1 && slab_ok(c, op.r[6], (c.B + p.SB - 1) / p.SB, g.Cin)) return VAE_EINVAL;
        a.slab = resolve(c, op.r[6], 4 * g.Cin, "conv.slab", &ok);
        if (bn_acc_ok(op, g.Cin))
            a.acc_save = resolve(c, op.r[7], (int64_t)8 * g.Cin + BNACC_FLOATS(g.Cin), "conv.in_bn+accumulators", &ok);
        if (!ok) return VAE_EINVAL;
    }
    return launch_igemm_gather(c, a, "igemm_gather_k(dgrad)");
}

// VAE_AMD_BNFIN=0: every virtual BatchNorm is finalised by a launch of its own (bnv_finalize_*_k), as before round 3
static int bnfin_enabled() {
    const char* e = getenv("VAE_AMD_BNFIN");
    return (e != nullptr && *e) ? atoi(e) != 0 : 1;
}
// The pending finalize as a launch of its own: the op behind the BatchNorm cannot do it in its prologue.
static int bn_flush_pending(const Ctx& c) {
    if (c.pend == 0) return VAE_OK;
    const BnFin& f = c.pend_fin;
    const int dir = c.pend;
    c.pend = 0;
    if (dir == 1) {
        hipLaunchKernelGGL(bnv_finalize_fwd_k, dim3(f.C), dim3(BNV_THREADS), 0, c.s, f.slab, f.nrows, f.gamma, f.beta, f.save,
                           f.rmean, f.rvar, f.C, f.eps, f.momentum, f.training, (float*)nullptr);
        VAE_TBYTES(16.0 * f.C * (f.nrows + 1));
        VAE_LAUNCH_CHECK("bnv_finalize_fwd_k");
    } else {
        hipLaunchKernelGGL(bnv_finalize_bwd_k, dim3(f.C), dim3(BNV_THREADS), 0, c.s, f.slab, f.nrows, f.save, f.dgamma, f.dbeta,
                           f.C, (int64_t)f.n_local, (float*)nullptr, (const float*)nullptr);
        VAE_TBYTES(16.0 * f.C * (f.nrows + 1));
        VAE_LAUNCH_CHECK("bnv_finalize_bwd_k");
    }
    return VAE_OK;
}
// Hands the pending finalize to a consumer kernel whose coefficient block is `coef` (the BatchNorm's save block).
// Returns the BnFin to pass (C = 0: none) and clears the pending state; a pending finalize of ANOTHER BatchNorm or
// direction is flushed first.
static int bn_take_pending(const Ctx& c, int dir, const float* coef, BnFin* out) {
    memset(out, 0, sizeof(*out));
    if (c.pend == 0) return VAE_OK;
    if (c.pend != dir || c.pend_fin.save != coef) return bn_flush_pending(c);
    *out = c.pend_fin;
    c.pend = 0;
    return VAE_OK;
}

static int bn_virtual_forward(const Ctx& c, const vae_op& op, int training) {
    const int C = op.d[0], HW = op.d[1];
    if (C < 1 || HW < 1) VAE_FAIL(VAE_EINVAL, "bn: bad shape C=%d HW=%d", C, HW);
    if (training && (int64_t)c.B * HW < 2) VAE_FAIL(VAE_EINVAL, "bn: training needs more than 1 value per channel");
    bool ok = true;
    const float* gamma = resolve(c, op.r[4], C, "bn.gamma", &ok);
    const float* beta = resolve(c, op.r[5], C, "bn.beta", &ok);
    float* save = resolve(c, op.r[6], (int64_t)8 * C, "bn.save", &ok);
    float* rm = resolve(c, op.r[7], C, "bn.running_mean", &ok);
    float* rv = resolve(c, op.r[8], C, "bn.running_var", &ok);
    const int nblocks = training ? c.last_blocks : 0;
    const bool dp = training && g_world > 1;
    // data-parallel: the ranks' rows are gathered right behind the local slab
    float* slab = resolve(c, op.r[9], (int64_t)4 * C * ((nblocks > 0 ? nblocks : 1) + (dp ? g_world : 0)), "bn.slab", &ok);
    if (!ok) return VAE_EINVAL;
    if (training && nblocks < 1) VAE_FAIL(VAE_EINVAL, "bn(virtual): no producer statistics");
    if (!training || bn_acc_ok(op, C)) {
        if (bn_acc_ok(op, C)) resolve(c, op.r[6], (int64_t)8 * C + BNACC_FLOATS(C), "bn.save+accumulators", &ok);
        if (!ok) return VAE_EINVAL;
    }
    if (!dp && C <= BNFIN_MAXC && bnfin_enabled() && (!training || bn_acc_ok(op, C))) {  // finalised by the consumer kernel's prologue (or by bn_flush_pending)
        BnFin& f = c.pend_fin;
        memset(&f, 0, sizeof(f));
        f.slab = slab; f.nrows = nblocks; f.C = C; f.training = training ? 1 : 0;
        f.gamma = gamma; f.beta = beta; f.save = save; f.rmean = rm; f.rvar = rv;
        f.eps = op.f[0]; f.momentum = op.f[1];
        c.pend = 1;
        c.last_blocks = 0;
        return VAE_OK;
    }
    const float* rows = slab;
    int nrows = nblocks;
    if (dp) {
        float* gath = slab + (int64_t)4 * C * nblocks;
        hipLaunchKernelGGL(bnv_finalize_fwd_k, dim3(C), dim3(BNV_THREADS), 0, c.s, slab, nblocks, gamma, beta, save, rm,
                           rv, C, op.f[0], op.f[1], training, gath + (int64_t)g_rank * 4 * C);
        VAE_LAUNCH_CHECK("bnv_finalize_fwd_k(local row)");
        if (g_allgather(g_allgather_user, gath, (int64_t)4 * C, (vae_stream_t)c.s) != 0)
            VAE_FAIL(VAE_EINVAL, "bn(virtual): the all-gather hook failed");
        rows = gath;
        nrows = g_world;
    }
    hipLaunchKernelGGL(bnv_finalize_fwd_k, dim3(C), dim3(BNV_THREADS), 0, c.s, rows, nrows, gamma, beta, save, rm, rv, C,
                       op.f[0], op.f[1], training, (float*)nullptr);
    VAE_TBYTES(16.0 * C * (nrows + 1));
    VAE_LAUNCH_CHECK("bnv_finalize_fwd_k");
    c.last_blocks = 0;
    return VAE_OK;
}
static int bn_virtual_backward(const Ctx& c, const vae_op& op) {
    const int C = op.d[0], HW = op.d[1];
    bool ok = true;
    float* dgamma = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[4]), C, "bn.dgamma", &ok);
    float* dbeta = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[5]), C, "bn.dbeta", &ok);
    float* save = resolve(c, op.r[6], (int64_t)8 * C, "bn.save", &ok);
    const int nblocks = c.last_blocks;
    const bool dp = g_world > 1;
    float* slab = resolve(c, op.r[9], (int64_t)4 * C * ((nblocks > 0 ? nblocks : 1) + (dp ? g_world : 0)), "bn.slab", &ok);
    if (!ok) return VAE_EINVAL;
    if (nblocks < 1) VAE_FAIL(VAE_EINVAL, "bn(virtual): no consumer gradient statistics");
    const int64_t n_local = (int64_t)c.B * HW;
    if (bn_acc_ok(op, C)) {
        resolve(c, op.r[6], (int64_t)8 * C + BNACC_FLOATS(C), "bn.save+accumulators", &ok);
        if (!ok) return VAE_EINVAL;
        BnFin& f = c.pend_fin;
        memset(&f, 0, sizeof(f));
        f.slab = slab; f.nrows = nblocks; f.C = C; f.training = 1;
        f.save = save; f.dgamma = dgamma; f.dbeta = dbeta; f.n_local = (float)n_local;
        c.pend = 2;
        c.last_blocks = 0;
        return VAE_OK;
    }
    if (dp) {
        float* gath = slab + (int64_t)4 * C * nblocks;
        float* mine = gath + (int64_t)g_rank * 4 * C;
        hipLaunchKernelGGL(bnv_finalize_bwd_k, dim3(C), dim3(BNV_THREADS), 0, c.s, slab, nblocks, save, dgamma, dbeta, C,
                           n_local, mine, (const float*)nullptr);
        VAE_LAUNCH_CHECK("bnv_finalize_bwd_k(local row)");
        if (g_allgather(g_allgather_user, gath, (int64_t)4 * C, (vae_stream_t)c.s) != 0)
            VAE_FAIL(VAE_EINVAL, "bn(virtual): the all-gather hook failed");
        hipLaunchKernelGGL(bnv_finalize_bwd_k, dim3(C), dim3(BNV_THREADS), 0, c.s, gath, g_world, save, dgamma, dbeta, C,
                           n_local, (float*)nullptr, (const float*)mine);
    } else {
        hipLaunchKernelGGL(bnv_finalize_bwd_k, dim3(C), dim3(BNV_THREADS), 0, c.s, slab, nblocks, save, dgamma, dbeta, C,
                           n_local, (float*)nullptr, (const float*)nullptr);
    }
    VAE_TBYTES(16.0 * C * (nblocks + 1));
    VAE_LAUNCH_CHECK("bnv_finalize_bwd_k");
    c.last_blocks = 0;
    return VAE_OK;
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
    float* save = resolve(c, op.r[6], (int64_t)8 * C, "bn.save", &ok);
    float* rm = resolve(c, op.r[7], C, "bn.running_mean", &ok);
    float* rv = resolve(c, op.r[8], C, "bn.running_var", &ok);
    float* part = resolve(c, op.r[9], (int64_t)2 * C * VAE_MAX_CHUNKS, "bn.partials", &ok);
    if (!ok) return VAE_EINVAL;
    if (training) {
        int bchunk, nchunks;
        if (chunking(c.B, C, HW, &bchunk, &nchunks) != VAE_OK) return VAE_EINVAL;
        hipLaunchKernelGGL(chan_reduce_k<CR_STATS>, dim3(C, nchunks), dim3(256), 0, c.s, x, nullptr, nullptr, nullptr,
                           part, c.B, C, HW, bchunk, 0, 0.f, 0);
        VAE_TNAME("chan_reduce_k<0>");
        VAE_TBYTES(4.0 * (double)total);
        VAE_LAUNCH_CHECK("chan_reduce_k<STATS>");
        if (g_world > 1) {  // data-parallel: one {mean, -, M2, n} row per rank behind the partials, then a merge
            float* gath = resolve(c, op.r[9] + (int64_t)2 * C * VAE_MAX_CHUNKS, (int64_t)4 * C * g_world, "bn.rows", &ok);
            if (!ok) return VAE_EINVAL;
            hipLaunchKernelGGL(bn_stats_finalize_k, dim3(C), dim3(64), 0, c.s, x, part, save, rm, rv, C, HW,
                               nchunks, (int64_t)c.B * HW, op.f[0], op.f[1], gath + (int64_t)g_rank * 4 * C);
            VAE_LAUNCH_CHECK("bn_stats_finalize_k(local row)");
            if (g_allgather(g_allgather_user, gath, (int64_t)4 * C, (vae_stream_t)c.s) != 0)
                VAE_FAIL(VAE_EINVAL, "bn: the all-gather hook failed");
            hipLaunchKernelGGL(bnv_finalize_fwd_k, dim3(C), dim3(BNV_THREADS), 0, c.s, gath, g_world, gamma, beta, save, rm,
                               rv, C, op.f[0], op.f[1], 1, (float*)nullptr);
            VAE_LAUNCH_CHECK("bnv_finalize_fwd_k(rows)");
        } else {
            hipLaunchKernelGGL(bn_stats_finalize_k, dim3(C), dim3(64), 0, c.s, x, part, save, rm, rv, C, HW,
                               nchunks, (int64_t)c.B * HW, op.f[0], op.f[1], (float*)nullptr);
            VAE_LAUNCH_CHECK("bn_stats_finalize_k");
        }
    } else {
        hipLaunchKernelGGL(bn_eval_prepare_k, dim3((C + 63) / 64), dim3(64), 0, c.s, rm, rv, save, C, op.f[0]);
        VAE_LAUNCH_CHECK("bn_eval_prepare_k");
    }
    hipLaunchKernelGGL(bn_apply_k, dim3(ew_grid(total)), dim3(256), 0, c.s, x, y, save, gamma, beta, total, C, HW,
                       (op.flags & VAE_F_BN_RELU) ? 1 : 0);
    VAE_TBYTES(8.0 * (double)total);
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
    float* save = resolve(c, op.r[6], (int64_t)8 * C, "bn.save", &ok);
    float* part = resolve(c, op.r[9], (int64_t)2 * C * VAE_MAX_CHUNKS, "bn.partials", &ok);
    if (!ok) return VAE_EINVAL;
    int bchunk, nchunks;
    if (chunking(c.B, C, HW, &bchunk, &nchunks) != VAE_OK) return VAE_EINVAL;
    hipLaunchKernelGGL(chan_reduce_k<CR_BNBWD>, dim3(C, nchunks), dim3(256), 0, c.s, x, dy, y, save, part, c.B, C, HW,
                       bchunk, 0, 0.f, (op.flags & VAE_F_BN_RELU) ? 1 : 0);
    VAE_TNAME("chan_reduce_k<1>");
    VAE_TBYTES(12.0 * (double)total);
    VAE_LAUNCH_CHECK("chan_reduce_k<BNBWD>");
    if (g_world > 1) {
        float* gath = resolve(c, op.r[9] + (int64_t)2 * C * VAE_MAX_CHUNKS, (int64_t)4 * C * g_world, "bn.rows", &ok);
        if (!ok) return VAE_EINVAL;
        float* mine = gath + (int64_t)g_rank * 4 * C;
        hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(C), dim3(64), 0, c.s, part, save, dgamma, dbeta, C,
                           nchunks, (int64_t)c.B * HW, mine);
        VAE_LAUNCH_CHECK("bn_bwd_finalize_k(local row)");
        if (g_allgather(g_allgather_user, gath, (int64_t)4 * C, (vae_stream_t)c.s) != 0)
            VAE_FAIL(VAE_EINVAL, "bn: the all-gather hook failed");
        hipLaunchKernelGGL(bn_bwd_rows_k, dim3((C + 63) / 64), dim3(64), 0, c.s, gath, g_world, mine, save, dgamma, dbeta, C);
        VAE_LAUNCH_CHECK("bn_bwd_rows_k");
    } else {
        hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(C), dim3(64), 0, c.s, part, save, dgamma, dbeta, C,
                           nchunks, (int64_t)c.B * HW, (float*)nullptr);
        VAE_LAUNCH_CHECK("bn_bwd_finalize_k");
    }
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_grid(total)), dim3(256), 0, c.s, x, dy, dx, save, gamma, total, C, HW);
    VAE_TBYTES(12.0 * (double)total);
    VAE_LAUNCH_CHECK("bn_bwd_apply_k");
    return VAE_OK;
}

// ---- linear ---------------------------------------------------------------------------------
// dynamic LDS beyond 64 KiB has to be granted per kernel, once per device
static int lin_ws_grant_lds() {
    static bool done[64];
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    VAE_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) VAE_FAIL(VAE_EINVAL, "device index %d out of range", dev);
    if (done[dev]) return VAE_OK;
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lin_ws_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done[dev] = true;
    return VAE_OK;
}
static int launch_lin(const Ctx& c, LinArgs a, const char* what) {
    if (a.B < 1 || a.N < 1 || a.K < 1) VAE_FAIL(VAE_EINVAL, "%s: bad shape %dx%dx%d", what, a.B, a.N, a.K);
    const int K16 = (a.K + 15) / 16 * 16;
    const int passes = (K16 + 319) / 320;
    a.KC = ((K16 + passes - 1) / passes + 15) / 16 * 16;
    a.Kp = a.KC + 1;
    a.d_KC = mk_fdiv((uint32_t)a.KC);
    a.d_KC4 = mk_fdiv((uint32_t)(a.KC / 4));
    // dwordx4 staging needs every row start 16-byte aligned and whole groups of 4 inside the edges
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    bool vec = a.K % 4 == 0 && a.lda % 4 == 0 && al16(a.A) && al16(a.W);
    if (a.A2 != nullptr) vec = vec && a.lda2 % 4 == 0 && al16(a.A2);
    if (a.wsn == 1) vec = vec && a.wsk % 4 == 0 && a.N % 4 == 0;  // W'[k][n], n contiguous
    else vec = vec && a.wsn % 4 == 0 && a.wsk == 1;               // W'[n][k], k contiguous
    if (a.nseg < 1) a.nseg = 1;
    if (a.nseg == 2) vec = vec && al16(a.Ab) && al16(a.Wb) && (a.A2b == nullptr || al16(a.A2b));
    int ny = (a.N + LIN_BN - 1) / LIN_BN;
    a.ny1 = 0;
    if (a.W2 != nullptr) {  // second forward head
        vec = vec && al16(a.W2) && (a.wsn != 1 || a.N2 % 4 == 0);
        a.ny1 = ny;
        ny += (a.N2 + LIN_BN - 1) / LIN_BN;
    }
    // large batches: the weights stay in LDS and the rows stream (lin.h: lin_ws_k)
    const int ws_min_b = [] { const char* e = getenv("VAE_LINWS_MIN_B"); return (e != nullptr && atoi(e) > 0) ? atoi(e) : 8192; }();  // (read at every launch: the tests switch it)
    if (!g_bf16 && vec && a.B >= ws_min_b) {
        LinWsPlan p;
        const int Kt = a.K * a.nseg;
        p.S4 = (((Kt >> 2) + 3) & ~3);         // k-steps rounded up to whole groups of four ...
        if (((p.S4 >> 2) & 1) == 0) p.S4 += 4;  // ... and 4 * odd: 16-byte operand reads without bank conflicts
        p.PX = (LWS_RT * p.S4 + 63) & ~63;
        const int budget = 40960 - 64 - 4 * p.PX;  // floats of the 160 KiB
        const int Nmax = a.W2 != nullptr && a.N2 > a.N ? a.N2 : a.N;
        bool fits = LWS_RT * (Kt >> 2) <= LWS_THREADS * LWS_NJ;
        p.NC = p.PW = 0;
        for (int g = 1; fits && g <= 16; ++g) {
            const int nc = ((Nmax + g - 1) / g + 15) & ~15;
            const int pw = (nc * p.S4 + 63) & ~63;
            if (4 * pw <= budget && nc <= 16 * LWS_WAVES * LWS_MAXNT) { p.NC = nc; p.PW = pw; break; }
        }
        if (fits && p.NC > 0) {
            int rc = lin_ws_grant_lds();
            if (rc) return rc;
            p.ng1 = (a.N + p.NC - 1) / p.NC;
            p.ng2 = a.W2 != nullptr ? (a.N2 + p.NC - 1) / p.NC : 0;
            const int rtiles = (a.B + LWS_RT - 1) / LWS_RT;
            int gx = 256 / (p.ng1 + p.ng2);  // one workgroup per CU: its weights fill the LDS
            if (gx < 1) gx = 1;
            if (gx > rtiles) gx = rtiles;
            const size_t wbytes = (size_t)4 * (p.PW + p.PX) * sizeof(float);
            hipLaunchKernelGGL(lin_ws_k, dim3(gx, p.ng1 + p.ng2), dim3(LWS_THREADS), wbytes, c.s, a, p);
            VAE_TNAME("lin_ws_k");
            VAE_TBYTES(4.0 * ((double)a.B * ((double)a.K * a.nseg * (a.A2 != nullptr ? 2 : 1) + a.N + a.N2) + (double)a.K * a.nseg * (a.N + a.N2)));
            VAE_LAUNCH_CHECK(what);
            return VAE_OK;
        }
    }
    const size_t bytes = (size_t)(LIN_BM * a.Kp + a.KC * LIN_WS) * sizeof(float);
    dim3 grid((a.B + LIN_BM - 1) / LIN_BM, ny);
    if (g_bf16) {
        if (vec) hipLaunchKernelGGL((lin_fwd_k<true, true>), grid, dim3(LIN_THREADS), bytes, c.s, a);
        else hipLaunchKernelGGL((lin_fwd_k<false, true>), grid, dim3(LIN_THREADS), bytes, c.s, a);
    } else {
        if (vec) hipLaunchKernelGGL((lin_fwd_k<true, false>), grid, dim3(LIN_THREADS), bytes, c.s, a);
        else hipLaunchKernelGGL((lin_fwd_k<false, false>), grid, dim3(LIN_THREADS), bytes, c.s, a);
    }
    VAE_TNAME("lin_fwd_k<%s, %s>", vec ? "true" : "false", g_bf16 ? "true" : "false");
    VAE_TBYTES(4.0 * ((double)a.B * ((double)a.K * a.nseg * (a.A2 != nullptr ? 2 : 1) + a.N + a.N2) + (double)a.K * a.nseg * (a.N + a.N2)));
    VAE_LAUNCH_CHECK(what);
    return VAE_OK;
}

// Batch slices of a Linear weight gradient: ~512 workgroups per layer (all layers share one launch); every slice
// owns a row of the layer's partial region.
struct LinWgPlan { int gx, gy, gz, chunks_per_block; };
// the whole-layer kernel (lin.h: lin_wgrad2_k) takes a layer whose smaller tile count is <= 7 and larger <= 32
static bool lin_wgrad2_shape_ok(int N, int K) {
    const int TN = (N + 15) / 16, TK = (K + 15) / 16;
    const int lo = TN < TK ? TN : TK, hi = TN < TK ? TK : TN;
    if (lo > LW2_FULLMAX || hi > 2 * LW2_WAVES) return false;
    const bool split_n = TN >= TK;
    const int tfc = lw2_fullc(lo);
    const int NS = lw2_stride(split_n ? N : 16 * tfc), KS = lw2_stride(split_n ? 16 * tfc : K);
    return (int64_t)LW2_RB * (NS + KS) <= 40000;  // floats of the 160 KiB
}
// ... from B = 8192: below, a slice is one 64-row chunk per workgroup whose 173 KB of rows and 115 KB partial row go
// through ONE CU's load / store path (~10 B/clk): 29.7 us against the tile kernel's 31.8 at B = 4096, and the finalize
// launch reads twice the rows.  At B = 65,536: 407 -> 254 us.  VAE_AMD_LINWG2 = 0 / 1 forces it off / on at any batch.
static bool lin_wgrad2_enabled(int B) {
    // (read at every call - the tests switch it; it must not change between sizing a program's partial regions and running it)
    const char* e = getenv("VAE_AMD_LINWG2");
    const int mode = (e != nullptr && *e) ? (atoi(e) != 0 ? 1 : 0) : -1;
    return mode < 0 ? B >= 8192 : mode == 1;
}
static LinWgPlan lin_wgrad_plan(int B, int N, int K) {
    LinWgPlan p;
    const int tiles = ((N + LW_BN - 1) / LW_BN) * ((K + 63) / 64);
    const int nch = (B + LW_BB - 1) / LW_BB;
    static const int target = [] { const char* e = getenv("VAE_LINWG_TARGET"); return (e != nullptr && atoi(e) > 0) ? atoi(e) : 512; }();
    int gx = (target + tiles - 1) / tiles;
    // (shapes the whole-layer kernel takes: one workgroup per slice, so the slices are what fills the GPU: one 64-row
    // chunk each up to 64 of them (128 from B = 32,768).  The tile kernel runs on the same slices when it has to, e.g. in bf16 mode)
    if (lin_wgrad2_enabled(B) && lin_wgrad2_shape_ok(N, K)) {
        // (measured: 64 / 128 slices give 0.7236 / 0.7343 ms per step at B = 8192, 1.2015 / 1.2062 at 16,384 - fewer rows
        // for the finalize launch - and 4.07 / 4.03 at 65,536, where the kernel itself takes 296 / 254 us)
        const char* e = getenv("VAE_LINWG2_SLICES");
        gx = (e != nullptr && atoi(e) > 0) ? atoi(e) : (nch >= 512 ? 128 : 64);
    }
    if (gx > nch) gx = nch;
    p.chunks_per_block = (nch + gx - 1) / gx;
    p.gx = (nch + p.chunks_per_block - 1) / p.chunks_per_block;
    p.gy = (N + LW_BN - 1) / LW_BN; p.gz = (K + 63) / 64;
    return p;
}
extern "C" int64_t vae_linear_wgrad_partial_floats(const vae_op* op, int B) {
    if (op == nullptr || op->kind != VAE_OP_LINEAR || B < 1 || op->d[0] < 1 || op->d[1] < 1) return 0;
    const int K = op->d[0], N = op->d[1];
    return (int64_t)lin_wgrad_plan(B, N, K).gx * ((int64_t)N * K + N);
}

// (Measured and dropped in round 3: this launch on a side stream / graph branch beside the encoder's convolution
// backward, forked behind the Linear run's input gradients and joined before wgrad_finalize_k: 0.508 -> 0.513 ms/step.
// Like the conv weight gradients in round 1, kernels that each fill the CUs only take turns.)
static int lin_wgrad2_grant_lds() {  // dynamic LDS beyond 64 KiB: once per device
    static bool done[64];
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    VAE_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) VAE_FAIL(VAE_EINVAL, "device index %d out of range", dev);
    if (done[dev]) return VAE_OK;
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lin_wgrad2_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done[dev] = true;
    return VAE_OK;
}
static int flush_linear_wgrads(const Ctx& c) {
    LinWgMulti& m = c.lin_wg;
    if (m.n == 0) return VAE_OK;
    int blocks = 0;
    bool v2 = !g_bf16 && m.n > 0 && lin_wgrad2_enabled(m.l[0].B);
    size_t lds2 = 0;
    for (int i = 0; i < m.n && v2; ++i) {
        const LinWgArgs& a = m.l[i];
        if (!lin_wgrad2_shape_ok(a.N, a.K)) { v2 = false; break; }
        const int TN = (a.N + 15) / 16, TK = (a.K + 15) / 16;
        const bool split_n = TN >= TK;
        const int tfc = lw2_fullc(split_n ? TK : TN);
        const size_t fl = (size_t)LW2_RB * (lw2_stride(split_n ? a.N : 16 * tfc) + lw2_stride(split_n ? 16 * tfc : a.K));
        if (fl > lds2) lds2 = fl;
    }
    if (v2) {  // one workgroup per (layer, batch slice)
        int rc = lin_wgrad2_grant_lds();
        if (rc) return rc;
        for (int i = 0; i < m.n; ++i) {
            m.l[i].first_block = blocks;
            blocks += m.l[i].gx;
        }
        hipLaunchKernelGGL(lin_wgrad2_k, dim3(blocks), dim3(LW2_THREADS), lds2 * sizeof(float), c.s, m);
    } else {
        for (int i = 0; i < m.n; ++i) {
            m.l[i].first_block = blocks;
            blocks += m.l[i].gx * m.l[i].gy * m.l[i].gz;
        }
        if (g_bf16) hipLaunchKernelGGL(lin_wgrad_k<true>, dim3(blocks), dim3(LIN_THREADS), 0, c.s, m);
        else hipLaunchKernelGGL(lin_wgrad_k<false>, dim3(blocks), dim3(LIN_THREADS), 0, c.s, m);
    }
    if (vae_tl_timer != nullptr) {
        double fl = 0;
        for (int i = 0; i < m.n; ++i)
            fl += (double)m.l[i].B * (m.l[i].K + (m.l[i].act != VAE_ACT_NONE ? 2.0 : 1.0) * m.l[i].N) + (double)m.l[i].gx * ((double)m.l[i].N * m.l[i].K + m.l[i].N);
        if (v2) VAE_TNAME("lin_wgrad2_k");
        else VAE_TNAME("lin_wgrad_k<%s>", g_bf16 ? "true" : "false");
        VAE_TBYTES(4.0 * fl);
    }
    m.n = 0;
    VAE_LAUNCH_CHECK("lin_wgrad_k");
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

// Two adjacent LINEAR ops that read the same input with the same shape of it (the mu / logvar heads, the two
// 288 -> 100 branches of the notebook encoder) run as one launch, forward (two heads) and in the input gradient
// (two K segments): every launch saved is >= 4.6 us of a step that is bound by them.
static bool linear_siblings(const vae_op& p, const vae_op& q) {
    return p.kind == VAE_OP_LINEAR && q.kind == VAE_OP_LINEAR && p.r[0] == q.r[0] && p.d[0] == q.d[0] &&
           p.d[2] == q.d[2] && p.act == q.act && p.f[0] == q.f[0];
}
// (input gradient: also the same destination and leading dimensions, and equally wide outputs)
static bool linear_dgrad_siblings(const vae_op& p, const vae_op& q) {
    return linear_siblings(p, q) && p.r[2] >= 0 && p.r[2] == q.r[2] && p.d[4] == q.d[4] && p.d[1] == q.d[1] &&
           p.d[3] == q.d[3] && p.d[5] == q.d[5];
}

static int linear_forward(const Ctx& c, const vae_op& op, const vae_op* sib = nullptr) {
    int rc = linear_check(c, op);
    if (rc) return rc;
    if (sib != nullptr && (rc = linear_check(c, *sib)) != VAE_OK) return rc;
    const int K = op.d[0], N = op.d[1], ldx = op.d[2], ldy = op.d[3];
    bool ok = true;
    LinArgs a;
    memset(&a, 0, sizeof(a));
    a.A = resolve(c, op.r[0], strided_extent(c.B, ldx, K), "linear.in", &ok);
    a.C = resolve(c, op.r[1], strided_extent(c.B, ldy, N), "linear.out", &ok);
    a.W = resolve(c, op.r[4], (int64_t)N * K, "linear.w", &ok);
    a.bias = resolve_opt(c, op.r[5], N, "linear.b", &ok);
    if (!ok) return VAE_EINVAL;
    a.B = c.B; a.N = N; a.K = K;
    a.lda = ldx; a.ldc = ldy; a.wsk = 1; a.wsn = K;
    a.act = op.act; a.slope = op.f[0];
    if (sib != nullptr) {
        a.N2 = sib->d[1]; a.ldc2 = sib->d[3];
        a.C2 = resolve(c, sib->r[1], strided_extent(c.B, a.ldc2, a.N2), "linear.out", &ok);
        a.W2 = resolve(c, sib->r[4], (int64_t)a.N2 * K, "linear.w", &ok);
        a.bias2 = resolve_opt(c, sib->r[5], a.N2, "linear.b", &ok);
        if (!ok) return VAE_EINVAL;
    }
    return launch_lin(c, a, "lin_fwd_k(fwd)");
}

// sib: the sibling processed just before (the later op of the pair), whose input gradient was left to this call;
// skip_dgrad: this op IS that later op.
static int linear_backward(const Ctx& c, const vae_op& op, const vae_op* sib = nullptr, bool skip_dgrad = false) {
    int rc = linear_check(c, op);
    if (rc) return rc;
    const int K = op.d[0], N = op.d[1], ldx = op.d[2], ldy = op.d[3], lddx = op.d[4], lddy = op.d[5];
    if (lddy < N || (op.r[2] >= 0 && lddx < K)) VAE_FAIL(VAE_EINVAL, "linear: bad grad leading dims");
    bool ok = true;
    const float* x = resolve(c, op.r[0], strided_extent(c.B, ldx, K), "linear.in", &ok);
    const float* y = resolve(c, op.r[1], strided_extent(c.B, ldy, N), "linear.out", &ok);
    float* dx = resolve_opt(c, op.r[2], strided_extent(c.B, lddx, K), "linear.din", &ok);
    const float* dy = resolve(c, op.r[3], strided_extent(c.B, lddy, N), "linear.dout", &ok);
    const float* w = resolve(c, op.r[4], (int64_t)N * K, "linear.w", &ok);
    float* dw = resolve(c, ((int64_t)2 << 56) | ref_off(op.r[4]), (int64_t)N * K, "linear.dw", &ok);
    float* db = op.r[5] < 0 ? nullptr : resolve(c, ((int64_t)2 << 56) | ref_off(op.r[5]), N, "linear.db", &ok);
    if (!ok) return VAE_EINVAL;

    // 1) dW[N][K] = (dY*act')^T X and db[N] = colsum(dY*act'): batch-chunked MFMA, one partial row per batch slice in the
    //    layer's own region, summed in a fixed order by wgrad_finalize_k (deferred launch; no float atomics)
    LinWgArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.dY = dy; wa.Y = y; wa.X = x; wa.dW = dw; wa.db = db;
    wa.lddy = lddy; wa.ldy = ldy; wa.ldx = ldx;
    wa.B = c.B; wa.N = N; wa.K = K; wa.act = op.act; wa.slope = op.f[0];
    const LinWgPlan lp = lin_wgrad_plan(c.B, N, K);
    wa.chunks_per_block = lp.chunks_per_block;
    wa.gx = lp.gx; wa.gy = lp.gy; wa.gz = lp.gz;
    {
        auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
        wa.vec = (N % 4 == 0 && K % 4 == 0 && N >= 4 && K >= 4 && lddy % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && al16(dy) && al16(y) &&
                  al16(x)) ? 1 : 0;
    }
    if (op.r[10] < 0) VAE_FAIL(VAE_EINVAL, "linear: backward needs the op's weight-gradient partial region (r[10], vae_linear_wgrad_partial_floats)");
    const int64_t prow = (int64_t)N * K + N;
    wa.part = resolve(c, op.r[10], lp.gx * prow, "linear.wgrad_partials", &ok);
    if (!ok) return VAE_EINVAL;
    // deferred: every Linear weight gradient of the pass goes into one launch (flush_linear_wgrads), and the fixed-order
    // sum of its per-slice rows into the conv layers' finalize launch (flush_wgrad_finalize)
    if (c.lin_wg.n == LIN_WG_MAX_LAYERS) {
        rc = flush_linear_wgrads(c);
        if (rc) return rc;
    }
    c.lin_wg.l[c.lin_wg.n++] = wa;
    if (c.wg_fin.n == WG_FIN_MAX_LAYERS) {
        rc = flush_wgrad_finalize(c);
        if (rc) return rc;
    }
    {
        WgFinLayer& F = c.wg_fin.l[c.wg_fin.n++];
        memset(&F, 0, sizeof(F));
        F.part = wa.part; F.dw = dw; F.dbias = db; F.nrows = lp.gx; F.NT = 1; F.T = 0;
        F.Sc = N; F.NK = K; F.C = N;
        F.plain = 1; F.stride = (int)prow; F.ncols = db != nullptr ? (int)prow : N * K;
    }
    // 2) dX[B][K] (=|+=) (dY*act') W
    if (dx != nullptr && !skip_dgrad) {
        LinArgs b;
        memset(&b, 0, sizeof(b));
        b.A = dy; b.lda = lddy;
        if (op.act != VAE_ACT_NONE) { b.A2 = y; b.lda2 = ldy; b.a_act = op.act; b.a_slope = op.f[0]; }
        b.W = w; b.wsk = K; b.wsn = 1;
        b.C = dx; b.ldc = lddx;
        b.B = c.B; b.N = K; b.K = N;
        b.accumulate = (op.flags & VAE_F_ACC_DIN) ? 1 : 0;
        if (sib != nullptr) {  // + (dY_sib*act') W_sib as a second K segment; the pair's first write decides = or +=
            b.nseg = 2;
            b.Ab = resolve(c, sib->r[3], strided_extent(c.B, lddy, N), "linear.dout", &ok);
            b.A2b = op.act != VAE_ACT_NONE ? resolve(c, sib->r[1], strided_extent(c.B, ldy, N), "linear.out", &ok) : nullptr;
            b.Wb = resolve(c, sib->r[4], (int64_t)N * K, "linear.w", &ok);
            if (!ok) return VAE_EINVAL;
            b.accumulate = (sib->flags & VAE_F_ACC_DIN) ? 1 : 0;
        }
        rc = launch_lin(c, b, "lin_fwd_k(dgrad)");
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
    VAE_TBYTES(16.0 * (double)n);
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
    VAE_TBYTES(24.0 * (double)n);
    VAE_LAUNCH_CHECK("reparam_bwd_k");
    return VAE_OK;
}

// ---- a run of LINEAR / REPARAM ops as one launch (fc.h) ---------------------------------------
#define FC_LDS_MAX_FLOATS 16384  // 64 KB of dynamic LDS
// Read once per run (make_ctx), so that a test can compare the two paths in one process.  VAE_AMD_FC=0: off, =1: on;
// unset: on up to FC_AUTO_MAX_BATCH samples.  The run is latency-bound at small batches (where one launch instead of
// twelve wins: 4096 samples 97 -> 62 us, 8192 samples 2 % of the step) and bound by the weights every 16-row workgroup
// streams from L2 at large ones (65,536 samples: 2.7 % of the step slower than op by op; break-even at 16,384).
#define FC_AUTO_MAX_BATCH 12288
static int fc_enabled(int B) {
    if (g_bf16) return 0;
    const char* e = getenv("VAE_AMD_FC");
    if (e != nullptr && e[0] != 0) return e[0] != '0';
    return B <= FC_AUTO_MAX_BATCH;
}
static inline bool fc_kind(const vae_op& op) { return op.kind == VAE_OP_LINEAR || op.kind == VAE_OP_REPARAM; }

struct FcBuf {
    int64_t ref;
    int width, ld, level;
    int stage, slot;  // producer (index into the unsorted stage list; slot 1: a REPARAM_BWD's second output)
    bool consumed;
};
struct FcBuild {
    FcChain ch;
    FcBuf buf[2 * FC_MAX_STAGES];
    int nbuf;
    bool overflow;
    FcBuild() : nbuf(0), overflow(false) { memset(&ch, 0, sizeof(ch)); }
    int find(int64_t ref, int width, int ld) const {
        for (int i = 0; i < nbuf; ++i)
            if (buf[i].ref == ref && buf[i].width == width && buf[i].ld == ld) return i;
        return -1;
    }
    // rows [B][width] at stride ld from `ref` share floats with a buffer of the run (column slices of one buffer, the
    // same buffer under another shape): such programs stay on the op-by-op path
    bool overlaps(int64_t ref, int width, int ld, int B) const {
        const int64_t lo = ref, hi = ref + (int64_t)(B - 1) * ld + width;  // (the space is in the top byte: no carry)
        for (int i = 0; i < nbuf; ++i) {
            const int64_t blo = buf[i].ref, bhi = blo + (int64_t)(B - 1) * buf[i].ld + buf[i].width;
            if (lo < bhi && blo < hi) return true;
        }
        return false;
    }
    int add_buf(int64_t ref, int width, int ld, int level, int stage, int slot) {
        if (nbuf == 2 * FC_MAX_STAGES) { overflow = true; return 0; }
        buf[nbuf] = FcBuf{ref, width, ld, level, stage, slot, false};
        return nbuf++;
    }
    FcStage* add_stage(int kind, int level) {
        if (ch.n == FC_MAX_STAGES) { overflow = true; return &ch.st[0]; }
        FcStage* s = &ch.st[ch.n++];
        memset(s, 0, sizeof(*s));
        s->kind = kind; s->level = level;
        s->in_off = s->in2_off = s->out_off = s->out2_off = -1;
        return s;
    }
    // an input of the run: the LDS rows of (ref, width, ld), loaded from global memory when nothing in the run made them
    int input(const Ctx& c, int64_t ref, int width, int ld, const char* what, bool* ok) {
        int b = find(ref, width, ld);
        if (b < 0) {
            if (overlaps(ref, width, ld, c.B)) { overflow = true; return 0; }
            FcStage* s = add_stage(FC_LOAD, 0);
            s->N = width;
            s->src = resolve(c, ref, (int64_t)(c.B - 1) * ld + width, what, ok);
            s->ld_src = ld;
            b = add_buf(ref, width, ld, 0, ch.n - 1, 0);
        }
        buf[b].consumed = true;
        return b;
    }
};

// LDS offsets for every buffer that a stage reads, stages sorted by level; 1: ready, 0: not for this path
static int fc_finish(FcBuild& fb, int B) {
    if (fb.overflow || fb.ch.n == 0) return 0;
    FcChain& ch = fb.ch;
    int off = 0, boff[2 * FC_MAX_STAGES], bstr[2 * FC_MAX_STAGES];
    for (int i = 0; i < fb.nbuf; ++i) {
        boff[i] = -1; bstr[i] = 0;
        if (!fb.buf[i].consumed) continue;
        bstr[i] = (fb.buf[i].width + 47) / 48 * 48 + 4;  // fc_chain_k reads A three k-steps (48 columns) at a time
        boff[i] = off;
        off += FC_ROWS * bstr[i];
    }
    if (off > FC_LDS_MAX_FLOATS) return 0;
    ch.lds_floats = off;
    for (int i = 0; i < fb.nbuf; ++i) {  // producers' destinations
        FcStage& s = ch.st[fb.buf[i].stage];
        if (fb.buf[i].slot == 0) { s.out_off = boff[i]; s.out_stride = bstr[i]; }
        else { s.out2_off = boff[i]; s.out2_stride = bstr[i]; }
    }
    for (int k = 0; k < ch.n; ++k) {  // consumers' sources: in_off / in2_off hold buffer indices + 1 until here
        FcStage& s = ch.st[k];
        if (s.kind == FC_LOAD) continue;
        const int b1 = s.in_off, b2 = s.in2_off;
        s.in_off = boff[b1]; s.in_stride = bstr[b1];
        if (b2 >= 0) { s.in2_off = boff[b2]; s.in2_stride = bstr[b2]; }
    }
    int nlev = 0;
    for (int k = 0; k < ch.n; ++k) nlev = ch.st[k].level + 1 > nlev ? ch.st[k].level + 1 : nlev;
    std::stable_sort(ch.st, ch.st + ch.n, [](const FcStage& a, const FcStage& b) { return a.level < b.level; });
    ch.B = B;
    // fc_chain_k addresses weights, outputs and act'(Y) rows as base + 32-bit byte offset, with 24-bit multiplies
    // (fc.h: fc_ld): anything outside those ranges goes to the per-op kernels
    for (int k = 0; k < ch.n; ++k) {
        const FcStage& s = ch.st[k];
        if (s.kind != FC_GEMM) continue;
        const int64_t lim = (int64_t)1 << 31, u24 = (int64_t)1 << 24;
        if (B >= u24 || s.N >= u24 || 4 * s.wsk >= u24 || 4 * s.wsn >= u24 || 4 * (int64_t)s.ldo >= u24 ||
            4 * (int64_t)s.ldmy >= u24)
            return 0;
        if (4 * ((s.N - 1) * s.wsn + (s.K - 1) * s.wsk + 4) >= lim || 4 * ((int64_t)(B - 1) * s.ldo + s.N) >= lim ||
            (s.mulY != nullptr && 4 * ((int64_t)(B - 1) * s.ldmy + s.N) >= lim))
            return 0;
    }
    for (int k = 0; k < ch.n; ++k) {  // fc_fetch loads the bias and act'(Y) unconditionally: absent ones read a weight
        FcStage& s = ch.st[k];
        if (s.kind != FC_GEMM) continue;
        s.has_bias = s.bias != nullptr; s.has_mul = s.mulY != nullptr;
        if (!s.has_bias) s.bias = s.W;
        if (!s.has_mul) s.mulY = s.W;
    }
    // every wave's work list: per level its share of the tiles (dealt round-robin), the element-wise stages, a barrier
    int cnt[FC_WAVES];
    for (int w = 0; w < FC_WAVES; ++w) cnt[w] = 0;
    auto push = [&](int w, int kind, int stage, int tile) {
        if (cnt[w] == FC_MAX_ITEMS) return false;
        const uint32_t it = (uint32_t)((kind << 12) | (stage << 8) | tile), sh = (cnt[w] & 1) * 16;
        ch.items[w][cnt[w] >> 1] = (ch.items[w][cnt[w] >> 1] & ~(0xffffu << sh)) | (it << sh);
        ++cnt[w];
        return true;
    };
    bool fits = true;
    for (int lev = 0, k = 0; lev < nlev; ++lev) {
        const int k0 = k;
        int g = 0;
        for (; k < ch.n && ch.st[k].level == lev; ++k) {
            if (ch.st[k].kind != FC_GEMM) continue;
            const int ntile = (ch.st[k].N + 15) / 16;
            if (ntile > 256) return 0;
            for (int t = 0; t < ntile; ++t, ++g) fits = fits && push(g % FC_WAVES, FC_IT_TILE, k, t);
        }
        for (int q = k0; q < k; ++q)
            if (ch.st[q].kind != FC_GEMM)
                for (int w = 0; w < FC_WAVES; ++w) fits = fits && push(w, FC_IT_ELEM, q, 0);
        for (int w = 0; w < FC_WAVES; ++w) fits = fits && push(w, FC_IT_BARRIER, 0, 0);
    }
    if (!fits) return 0;
    for (int w = 0; w < FC_WAVES; ++w) ch.n_items[w] = (uint32_t)cnt[w];
    return 1;
}
static int fc_launch(const Ctx& c, const FcChain& ch, const char* what) {
    bool vec = false;  // some GEMM stage takes dwordx4 weight loads
    for (int k = 0; k < ch.n; ++k) vec = vec || (ch.st[k].kind == FC_GEMM && ch.st[k].vec);
    const dim3 grid((c.B + FC_ROWS - 1) / FC_ROWS);
    const size_t bytes = (size_t)ch.lds_floats * sizeof(float);
    if (vec) hipLaunchKernelGGL(fc_chain_k<true>, grid, dim3(FC_THREADS), bytes, c.s, ch);
    else hipLaunchKernelGGL(fc_chain_k<false>, grid, dim3(FC_THREADS), bytes, c.s, ch);
    if (vae_tl_timer != nullptr) {  // rows loaded and stored once per sample, every weight matrix once
        double fl = 0;
        for (int k = 0; k < ch.n; ++k) {
            const FcStage& st = ch.st[k];
            if (st.kind == FC_LOAD) fl += (double)c.B * st.N;
            else if (st.kind == FC_GEMM) fl += (double)c.B * st.N * (st.has_mul ? 2 : 1) + (double)st.K * st.N * (st.W2 != nullptr ? 2 : 1);
            else if (st.kind == FC_REPARAM_FWD) fl += 2.0 * c.B * st.N;
            else fl += 5.0 * c.B * st.N;  // REPARAM_BWD: mu, logvar, eps in, two gradients out
        }
        VAE_TNAME("fc_chain_k<%s>", vec ? "true" : "false");
        VAE_TBYTES(4.0 * fl);
    }
    VAE_LAUNCH_CHECK(what);
    return VAE_OK;
}

// forward of ops[0..n) (all LINEAR / REPARAM).  1: launched, 0: not for this path (run the ops one by one), < 0: error
static int fc_forward(const Ctx& c, const vae_op* ops, int n) {
    FcBuild fb;
    bool ok = true;
    for (int j = 0; j < n && !fb.overflow; ++j) {
        const vae_op& op = ops[j];
        if (op.kind == VAE_OP_LINEAR) {
            if (linear_check(c, op) != VAE_OK) return VAE_EINVAL;
            const int K = op.d[0], N = op.d[1], ldx = op.d[2], ldy = op.d[3];
            const int bi = fb.input(c, op.r[0], K, ldx, "linear.in", &ok);
            if (fb.overlaps(op.r[1], N, ldy, c.B)) return 0;
            FcStage* s = fb.add_stage(FC_GEMM, fb.buf[bi].level + 1);
            s->in_off = bi;
            s->K = K; s->N = N;
            s->W = resolve(c, op.r[4], (int64_t)N * K, "linear.w", &ok);
            s->wsk = 1; s->wsn = K;
            s->vec = (K % 4 == 0 && ((uintptr_t)s->W & 15) == 0) ? 1 : 0;
            s->bias = resolve_opt(c, op.r[5], N, "linear.b", &ok);
            s->act = op.act; s->slope = op.f[0];
            s->out = resolve(c, op.r[1], strided_extent(c.B, ldy, N), "linear.out", &ok);
            s->ldo = ldy;
            fb.add_buf(op.r[1], N, ldy, s->level, fb.ch.n - 1, 0);
        } else {
            const int L = op.d[0];
            if (L < 1) VAE_FAIL(VAE_EINVAL, "reparam: bad latent");
            const int bm = fb.input(c, op.r[0], L, L, "reparam.mu", &ok);
            const int bl = fb.input(c, op.r[1], L, L, "reparam.logvar", &ok);
            if (fb.overlaps(op.r[3], L, L, c.B)) return 0;
            const int lev = (fb.buf[bm].level > fb.buf[bl].level ? fb.buf[bm].level : fb.buf[bl].level) + 1;
            FcStage* s = fb.add_stage(FC_REPARAM_FWD, lev);
            s->in_off = bm; s->in2_off = bl;
            s->N = L;
            s->src = resolve(c, op.r[2], (int64_t)c.B * L, "reparam.eps", &ok);
            s->ld_src = L;
            s->out = resolve(c, op.r[3], (int64_t)c.B * L, "reparam.z", &ok);
            s->ldo = L;
            fb.add_buf(op.r[3], L, L, lev, fb.ch.n - 1, 0);
        }
        if (!ok) return VAE_EINVAL;
    }
    if (!fc_finish(fb, c.B)) return 0;
    return fc_launch(c, fb.ch, "fc_chain_k(fwd)") == VAE_OK ? 1 : VAE_EHIP;
}

// input gradients of ops[0..n) (walked backwards); the weight gradients are registered by the caller afterwards
static int fc_backward(const Ctx& c, const vae_op* ops, int n) {
    FcBuild fb;
    bool ok = true;
    // the LDS rows of an op's dout, times act'(Y) of that op
    auto dout_rows = [&](const vae_op& op) -> int {
        const int N = op.d[1], ldy = op.d[3], lddy = op.d[5];
        const int before = fb.nbuf;
        const int b = fb.input(c, op.r[3], N, lddy, "linear.dout", &ok);
        if (fb.overflow) return 0;
        FcStage& p = fb.ch.st[fb.buf[b].stage];
        if (b < before && (p.mulY != nullptr || fb.buf[b].slot != 0 || p.kind == FC_REPARAM_BWD) && op.act != VAE_ACT_NONE) {
            fb.overflow = true;  // a second consumer, or a producer without the multiplier
            return 0;
        }
        if (op.act != VAE_ACT_NONE) {
            p.mulY = resolve(c, op.r[1], strided_extent(c.B, ldy, N), "linear.out", &ok);
            p.ldmy = ldy; p.mul_act = op.act; p.mul_slope = op.f[0];
        }
        return b;
    };
    for (int j = n - 1; j >= 0 && !fb.overflow; --j) {
        const vae_op& op = ops[j];
        if (op.kind == VAE_OP_LINEAR) {
            if (linear_check(c, op) != VAE_OK) return VAE_EINVAL;
            const int K = op.d[0], N = op.d[1], lddx = op.d[4], lddy = op.d[5];
            if (lddy < N || (op.r[2] >= 0 && lddx < K)) VAE_FAIL(VAE_EINVAL, "linear: bad grad leading dims");
            if (op.r[2] < 0) continue;  // no input gradient wanted
            const vae_op* sib = (j - 1 >= 0 && linear_dgrad_siblings(ops[j - 1], op)) ? &ops[j - 1] : nullptr;
            const int b1 = dout_rows(op);
            const int b2 = sib != nullptr ? dout_rows(*sib) : -1;
            if (fb.overflow) break;
            if (fb.overlaps(op.r[2], K, lddx, c.B)) return 0;  // a second writer of one gradient buffer inside the run
            int lev = fb.buf[b1].level;
            if (b2 >= 0 && fb.buf[b2].level > lev) lev = fb.buf[b2].level;
            FcStage* s = fb.add_stage(FC_GEMM, lev + 1);
            s->in_off = b1; s->in2_off = b2;
            s->K = N; s->N = K;
            s->W = resolve(c, op.r[4], (int64_t)N * K, "linear.w", &ok);
            if (sib != nullptr) s->W2 = resolve(c, sib->r[4], (int64_t)N * K, "linear.w", &ok);
            s->wsk = K; s->wsn = 1;
            s->out = resolve(c, op.r[2], strided_extent(c.B, lddx, K), "linear.din", &ok);
            s->ldo = lddx;
            s->accumulate = (op.flags & VAE_F_ACC_DIN) ? 1 : 0;  // (of a pair: the later op's, which writes first)
            fb.add_buf(op.r[2], K, lddx, s->level, fb.ch.n - 1, 0);
            if (sib != nullptr) --j;
        } else {
            const int L = op.d[0];
            const int64_t cnt = (int64_t)c.B * L;
            const int bz = fb.input(c, op.r[4], L, L, "reparam.dz", &ok);
            if (fb.overflow) break;
            if (fb.overlaps(op.r[5], L, L, c.B) || fb.overlaps(op.r[6], L, L, c.B)) return 0;
            FcStage* s = fb.add_stage(FC_REPARAM_BWD, fb.buf[bz].level + 1);
            s->in_off = bz;
            s->N = L;
            s->mu = resolve(c, op.r[0], cnt, "reparam.mu", &ok);
            s->lv = resolve(c, op.r[1], cnt, "reparam.logvar", &ok);
            s->src = resolve(c, op.r[2], cnt, "reparam.eps", &ok);
            s->ld_src = L;
            s->out = resolve(c, op.r[5], cnt, "reparam.dmu", &ok);
            s->out2 = resolve(c, op.r[6], cnt, "reparam.dlogvar", &ok);
            s->ldo = L;
            if (op.flags & VAE_F_EXT_GRAD) {
                s->g_mu = resolve(c, op.r[7], cnt, "reparam.g_mu", &ok);
                s->g_lv = resolve(c, op.r[8], cnt, "reparam.g_logvar", &ok);
            }
            if (op.flags & VAE_F_KL_GRAD) s->gscale = resolve_opt(c, op.r[9], 1, "reparam.gscale", &ok);
            s->kl_grad = (op.flags & VAE_F_KL_GRAD) ? 1 : 0;
            fb.add_buf(op.r[5], L, L, s->level, fb.ch.n - 1, 0);
            fb.add_buf(op.r[6], L, L, s->level, fb.ch.n - 1, 1);
        }
        if (!ok) return VAE_EINVAL;
    }
    if (!fc_finish(fb, c.B)) return 0;
    return fc_launch(c, fb.ch, "fc_chain_k(bwd)") == VAE_OK ? 1 : VAE_EHIP;
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
    VAE_TBYTES(8.0 * (double)rows);
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
    VAE_TBYTES(4.0 * (double)rows * (L + 1));
    VAE_LAUNCH_CHECK("slice0_bwd_k");
    return VAE_OK;
}

static int elbo_fwd_impl(hipStream_t s, const float* xr, const float* tg, const float* mu, const float* lv, float* out,
                         float* partials, int B, int nps, int64_t t_sb, int64_t t_se, int latent, int bce,
                         int accumulate, bool finalize = true) {
    hipLaunchKernelGGL(elbo_fwd_k, dim3(VAE_ELBO_BLOCKS), dim3(256), 0, s, xr, tg, mu, lv, partials, (int64_t)B, nps,
                       t_sb, t_se, latent, bce);
    VAE_TBYTES(8.0 * (double)B * (nps + latent));
    VAE_LAUNCH_CHECK("elbo_fwd_k");
    if (!finalize) return VAE_OK;  // VAE_F_LOSS_IN_BWD: summed by the op's backward launch
    hipLaunchKernelGGL(elbo_finalize_k, dim3(1), dim3(256), 0, s, partials, out, VAE_ELBO_BLOCKS, accumulate);
    VAE_LAUNCH_CHECK("elbo_finalize_k");
    return VAE_OK;
}
static inline int64_t target_extent(int B, int nps, int64_t sb, int64_t se) {
    return (int64_t)(B - 1) * sb + (int64_t)(nps - 1) * se + 1;
}
// VAE_F_LOSS_IN_BWD promises that the op's backward runs behind every forward: with the target laid out like x_rec the
// loss terms and the gradient then come from one pass over x_rec (elem.h: elbo_fused_k; VAE_AMD_ELBO_FUSED=0: the
// two-launch form); the fixed-order sum of the workgroups' pairs rides on the weight-gradient reduction at the end of
// the backward call that holds the op.
static bool elbo_fused(const vae_op& op) {
    static const int on = [] { const char* e = getenv("VAE_AMD_ELBO_FUSED"); return (e != nullptr && *e) ? atoi(e) != 0 : 1; }();
    return on && (op.flags & VAE_F_LOSS_IN_BWD) && op.d[2] == 1 && op.d[1] == op.d[0];
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
    if (elbo_fused(op)) return VAE_OK;  // the op's backward launch computes the loss as well (elbo_fused_k)
    return elbo_fwd_impl(c.s, xr, tg, mu, lv, out, part, c.B, nps, op.d[1], op.d[2], latent,
                         (op.flags & VAE_F_BCE) ? 1 : 0, op.d[5], !(op.flags & VAE_F_LOSS_IN_BWD));
}
static int elbo_backward(const Ctx& c, const vae_op& op, float* zero_buf = nullptr, int64_t zero_n = 0,
                         const ZeroRanges* zr = nullptr) {
    const int nps = op.d[0], latent = op.d[3];
    bool ok = true;
    const float* xr = resolve(c, op.r[0], (int64_t)c.B * nps, "elbo.x_rec", &ok);
    const float* tg = resolve(c, op.r[1], target_extent(c.B, nps, op.d[1], op.d[2]), "elbo.target", &ok);
    float* dxr = resolve(c, op.r[5], (int64_t)c.B * nps, "elbo.dx_rec", &ok);
    const float* gs = resolve_opt(c, op.r[7], 1, "elbo.gscale", &ok);
    const float* fin_part = nullptr;
    float* fin_out = nullptr;
    if (op.flags & VAE_F_LOSS_IN_BWD) {  // the forward left the sum of its partials to this launch
        fin_part = resolve(c, op.r[6], 2 * VAE_ELBO_BLOCKS, "elbo.partials", &ok);
        fin_out = resolve(c, op.r[4], 4, "elbo.out", &ok);
    }
    if (!ok) return VAE_EINVAL;
    if (elbo_fused(op)) {
        const float* mu = resolve(c, op.r[2], (int64_t)c.B * latent, "elbo.mu", &ok);
        const float* lv = resolve(c, op.r[3], (int64_t)c.B * latent, "elbo.logvar", &ok);
        float* part = resolve(c, op.r[6], 2 * VAE_ELBO_BLOCKS, "elbo.partials", &ok);
        float* out = resolve(c, op.r[4], 4, "elbo.out", &ok);
        if (!ok) return VAE_EINVAL;
        hipLaunchKernelGGL(elbo_fused_k, dim3(VAE_ELBO_BLOCKS), dim3(256), 0, c.s, xr, tg, mu, lv, gs, dxr, part,
                           (int64_t)c.B, nps, latent, (op.flags & VAE_F_BCE) ? 1 : 0, zero_buf, zero_n,
                           zr != nullptr ? *zr : ZeroRanges{});
        // the sum of the pairs rides on this pass's weight-gradient reduction (flush_wgrad_finalize)
        c.wg_fin.loss_part = part; c.wg_fin.loss_out = out; c.wg_fin.loss_blocks = VAE_ELBO_BLOCKS;
        c.wg_fin.loss_accumulate = op.d[5];
        VAE_TNAME("elbo_fused_k");
        VAE_TBYTES(12.0 * (double)c.B * nps + 8.0 * (double)c.B * latent + 4.0 * (double)zero_n);
        VAE_LAUNCH_CHECK("elbo_fused_k");
        return VAE_OK;
    }
    // KL gradients are folded into the REPARAM op's backward (VAE_F_KL_GRAD)
    hipLaunchKernelGGL(elbo_bwd_k, dim3(ew_grid((int64_t)c.B * nps)), dim3(256), 0, c.s, xr, tg, nullptr, nullptr, gs,
                       dxr, nullptr, nullptr, (int64_t)c.B, nps, (int64_t)op.d[1], (int64_t)op.d[2], latent,
                       (op.flags & VAE_F_BCE) ? 1 : 0, zero_buf, zero_n, fin_part, fin_out, VAE_ELBO_BLOCKS, op.d[5],
                       zr != nullptr ? *zr : ZeroRanges{});
    VAE_TBYTES(12.0 * (double)c.B * nps + 4.0 * (double)zero_n);
    VAE_LAUNCH_CHECK("elbo_bwd_k");
    return VAE_OK;
}

#include "seq1d_host.h"

// ---- executor --------------------------------------------------------------------------------
static int make_ctx(Ctx* c, const vae_op* ops, int n_ops, float* const* bases, const int64_t* sizes, int n_bases,
                    int B, vae_stream_t stream) {
    if (ops == nullptr || n_ops < 0 || bases == nullptr || sizes == nullptr) VAE_FAIL(VAE_EINVAL, "null argument");
    if (n_bases < 4 || n_bases > 128) VAE_FAIL(VAE_EINVAL, "n_bases=%d out of range", n_bases);
    if (B < 1 || B > (1 << 24)) VAE_FAIL(VAE_EINVAL, "batch %d out of range", B);
    c->bases = bases; c->sizes = sizes; c->n_bases = n_bases; c->B = B; c->s = (hipStream_t)stream;
    c->fc = fc_enabled(B);
    c->last_blocks = 0;
    c->lin_wg.n = 0;
    c->wg_fin.n = 0;
    c->wg_fin.loss_part = nullptr; c->wg_fin.loss_out = nullptr; c->wg_fin.loss_blocks = 0; c->wg_fin.loss_accumulate = 0;
    c->pend = 0;
    memset(&c->pend_fin, 0, sizeof(c->pend_fin));
    return VAE_OK;
}

// Clears the save blocks + accumulators of every BatchNorm flagged VAE_F_BN_ACC: one memset when they lie back to back
// (each padded to 64 floats: what vae_amd/program.py allocates), else one per block.
// zr != nullptr: instead, return the accumulator ranges (not the save blocks: the backward pass still reads them) for a
// kernel that clears them on its way (elbo_bwd_k); more than ZERO_RANGES_MAX blocks are cleared by memsets here.
static int bn_zero_accumulators(const Ctx& c, const vae_op* ops, int n_ops, ZeroRanges* zr = nullptr) {
    if (zr != nullptr) zr->cnt = 0;
    if (g_world > 1 || !bnfin_enabled()) return VAE_OK;
    int64_t lo = -1, hi = -1, sum = 0;
    int nb = 0;
    bool ok = true;
    for (int i = 0; i < n_ops; ++i) {
        const vae_op& op = ops[i];
        if (op.kind != VAE_OP_BN || !(op.flags & VAE_F_BN_ACC) || op.d[0] < 1 || op.d[0] > BNFIN_MAXC) continue;
        const int64_t n = (int64_t)8 * op.d[0] + BNACC_FLOATS(op.d[0]), npad = (n + 63) / 64 * 64;
        if (ref_space(op.r[6]) != 0) VAE_FAIL(VAE_EINVAL, "bn: accumulator blocks live in the WORK space");
        resolve(c, op.r[6], n, "bn.save+accumulators", &ok);
        if (!ok) return VAE_EINVAL;
        const int64_t off = ref_off(op.r[6]);
        if (zr != nullptr) {
            if (zr->cnt < ZERO_RANGES_MAX) {
                zr->p[zr->cnt] = c.bases[0] + off + 8 * op.d[0];
                zr->n[zr->cnt] = BNACC_FLOATS(op.d[0]);
                ++zr->cnt;
            } else {
                VAE_HIP(hipMemsetAsync(c.bases[0] + off + 8 * op.d[0], 0, (size_t)BNACC_FLOATS(op.d[0]) * sizeof(float), c.s));
            }
            continue;
        }
        if (lo < 0 || off < lo) lo = off;
        if (off + n > hi) hi = off + n;
        sum += npad;
        ++nb;
    }
    if (nb == 0) return VAE_OK;
    if ((hi - lo + 63) / 64 * 64 == sum) {  // (at the start of a forward pass the save blocks are dead as well)
        VAE_HIP(hipMemsetAsync(c.bases[0] + lo, 0, (size_t)(hi - lo) * sizeof(float), c.s));
        return VAE_OK;
    }
    for (int i = 0; i < n_ops; ++i) {
        const vae_op& op = ops[i];
        if (op.kind != VAE_OP_BN || !(op.flags & VAE_F_BN_ACC) || op.d[0] < 1 || op.d[0] > BNFIN_MAXC) continue;
        VAE_HIP(hipMemsetAsync(c.bases[0] + ref_off(op.r[6]), 0, (size_t)(8 * op.d[0] + BNACC_FLOATS(op.d[0])) * sizeof(float), c.s));
    }
    return VAE_OK;
}
static inline bool fused_conv_op(const vae_op& op) {
    return (op.kind == VAE_OP_CONV || op.kind == VAE_OP_CONVT) && (op.flags & VAE_F_FUSED);
}
static int forward_op(const Ctx& c, const vae_op& op, int i, int training) {
    if (c.pend && !fused_conv_op(op)) {
        const int rc = bn_flush_pending(c);
        if (rc) return rc;
    }
    switch (op.kind) {
        case VAE_OP_CONV:
        case VAE_OP_CONVT: return (op.flags & VAE_F_FUSED) ? conv_forward_fused(c, op, training) : conv_forward(c, op);
        case VAE_OP_BN: return (op.flags & VAE_F_FUSED) ? bn_virtual_forward(c, op, training) : bn_forward(c, op, training);
        case VAE_OP_LINEAR: return linear_forward(c, op);
        case VAE_OP_REPARAM: return reparam_forward(c, op);
        case VAE_OP_SLICE0: return slice0_forward(c, op);
        case VAE_OP_ELBO: return elbo_forward(c, op);
        default: vae_set_error("op %d: unknown kind %d", i, op.kind); return VAE_EINVAL;
    }
}

extern "C" int vae_run_forward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                               int n_bases, int B, int training, vae_stream_t stream) {
    Ctx c;
    int rc = make_ctx(&c, ops, n_ops, bases, base_sizes, n_bases, B, stream);
    if (rc) return rc;
    {   // a whole 1-D VAE program runs as one launch per BatchNorm-delimited phase (seq1d.h)
        static thread_local SqProgram sq;
        if (sq_use(ops, n_ops, B, &sq)) return sq_forward(c, ops, &sq, training);
    }
    // training == 2: the caller vouches that the BatchNorm accumulators are zero already (a fused train step: the
    // workspace starts zeroed and every backward pass that begins with the ELBO op clears them again)
    if (training == 1 && (rc = bn_zero_accumulators(c, ops, n_ops)) != VAE_OK) return rc;
    for (int i = 0; i < n_ops && rc == VAE_OK; ++i) {
        if (c.pend && !fused_conv_op(ops[i]) && (rc = bn_flush_pending(c)) != VAE_OK) break;
        if (c.fc && fc_kind(ops[i])) {  // a run of Linear / reparam ops: one launch (fc.h)
            int n = 1;
            while (i + n < n_ops && fc_kind(ops[i + n])) ++n;
            const int r = n >= 2 ? fc_forward(c, ops + i, n) : 0;
            if (r < 0) return r;
            if (r == 1) { i += n - 1; continue; }
        }
        if (i + 1 < n_ops && linear_siblings(ops[i], ops[i + 1])) {
            rc = linear_forward(c, ops[i], &ops[i + 1]);
            ++i;
        } else {
            rc = forward_op(c, ops[i], i, training);
        }
    }
    if (rc == VAE_OK) rc = bn_flush_pending(c);
    return rc;
}

// Same launches as vae_run_forward with a HIP event between consecutive ops ON THE LAUNCH STREAM; fills
// op_ms[i] with the device time of op i.  Synchronises the stream; not capturable.
extern "C" int vae_time_forward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                                int n_bases, int B, int training, vae_stream_t stream, float* op_ms) {
    if (op_ms == nullptr || n_ops < 1 || n_ops > 4096) VAE_FAIL(VAE_EINVAL, "time_forward: bad arguments");
    Ctx c;
    int rc = make_ctx(&c, ops, n_ops, bases, base_sizes, n_bases, B, stream);
    if (rc) return rc;
    hipEvent_t* ev = new hipEvent_t[n_ops + 1];
    int made = 0;
    for (; made <= n_ops; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) break;
    if (made <= n_ops) rc = VAE_EHIP, vae_set_error("time_forward: hipEventCreate failed");
    if (rc == VAE_OK && training) rc = bn_zero_accumulators(c, ops, n_ops);
    for (int i = 0; i < n_ops && rc == VAE_OK; ++i) {
        if (hipEventRecord(ev[i], c.s) != hipSuccess) { rc = VAE_EHIP; vae_set_error("hipEventRecord failed"); break; }
        rc = forward_op(c, ops[i], i, training);
    }
    if (rc == VAE_OK) rc = bn_flush_pending(c);
    if (rc == VAE_OK && (hipEventRecord(ev[n_ops], c.s) != hipSuccess || hipEventSynchronize(ev[n_ops]) != hipSuccess))
        rc = VAE_EHIP, vae_set_error("time_forward: event record/synchronize failed");
    for (int i = 0; i < n_ops && rc == VAE_OK; ++i)
        if (hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]) != hipSuccess)
            rc = VAE_EHIP, vae_set_error("hipEventElapsedTime failed");
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    return rc;
}

// The fused 1-D path's forward + backward with a HIP event before every launch ON THE LAUNCH STREAM: ms[k], bytes[k]
// (algorithmic bytes of launch k) and tags (8 chars per launch: fe0.., fh, fd1.., elbofin, bd.., bh, be.., reduce).
// Synchronises the stream; not capturable.  VAE_EINVAL when the program is not a seq1d program.
extern "C" int vae_time_seq(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes, int n_bases,
                            int B, vae_stream_t stream, float* ms, double* bytes, char* tags, int cap, int* n_launches) {
    if (ms == nullptr || bytes == nullptr || tags == nullptr || n_launches == nullptr || cap < 1)
        VAE_FAIL(VAE_EINVAL, "time_seq: bad arguments");
    Ctx c;
    int rc = make_ctx(&c, ops, n_ops, bases, base_sizes, n_bases, B, stream);
    if (rc) return rc;
    if (bases[2] == nullptr) VAE_FAIL(VAE_EINVAL, "time_seq needs the GRAD space (2) bound");
    static thread_local SqProgram sq;
    if (!sq_use(ops, n_ops, B, &sq)) VAE_FAIL(VAE_EINVAL, "time_seq: not a fused 1-D (seq1d) program");
    SqTimer* tm = new SqTimer;
    tm->n = 0; tm->failed = false;
    int made = 0;
    for (; made <= SQ_MAX_MARKS; ++made)
        if (hipEventCreate(&tm->ev[made]) != hipSuccess) break;
    if (made <= SQ_MAX_MARKS) rc = VAE_EHIP, vae_set_error("time_seq: hipEventCreate failed");
    if (rc == VAE_OK) rc = sq_forward(c, ops, &sq, 1, tm);
    if (rc == VAE_OK) rc = sq_backward(c, ops, &sq, tm);
    if (rc == VAE_OK && (tm->failed || hipEventRecord(tm->ev[tm->n], c.s) != hipSuccess ||
                         hipEventSynchronize(tm->ev[tm->n]) != hipSuccess))
        rc = VAE_EHIP, vae_set_error("time_seq: event record/synchronize failed");
    if (rc == VAE_OK && tm->n > cap) rc = VAE_EINVAL, vae_set_error("time_seq: %d launches, room for %d", tm->n, cap);
    for (int k = 0; k < tm->n && rc == VAE_OK; ++k) {
        if (hipEventElapsedTime(&ms[k], tm->ev[k], tm->ev[k + 1]) != hipSuccess)
            rc = VAE_EHIP, vae_set_error("hipEventElapsedTime failed");
        bytes[k] = tm->bytes[k];
        memcpy(tags + 8 * k, tm->tag[k], 8);
    }
    if (rc == VAE_OK) *n_launches = tm->n;
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(tm->ev[i]);
    delete tm;
    return rc;
}

extern "C" int vae_run_backward_range(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                                      int n_bases, int B, vae_stream_t stream, int op_begin, int op_end, int zero_grads) {
    Ctx c;
    int rc = make_ctx(&c, ops, n_ops, bases, base_sizes, n_bases, B, stream);
    if (rc) return rc;
    if (bases[2] == nullptr) VAE_FAIL(VAE_EINVAL, "backward needs the GRAD space (2) bound");
    if (op_begin < 0 || op_end > n_ops || op_begin > op_end)
        VAE_FAIL(VAE_EINVAL, "backward: op range [%d, %d) outside the program of %d ops", op_begin, op_end, n_ops);
    {
        static thread_local SqProgram sq;
        if (sq_use(ops, n_ops, B, &sq)) {
            if (op_begin != 0 || op_end != n_ops)
                VAE_FAIL(VAE_EINVAL, "backward: a fused 1-D program (seq1d) runs whole, not in op ranges");
            return sq_backward(c, ops, &sq);
        }
    }
    // gradients of parameters that several ops share accumulate, and the direct fallback kernels (conv.h) add with float
    // atomics: start from zero (cleared by the ELBO op's kernel when the range ends with one: it runs first and
    // nothing before it touches the GRAD space)
    const bool zero_in_elbo = zero_grads && op_end > op_begin && ops[op_end - 1].kind == VAE_OP_ELBO;
    if (zero_grads && !zero_in_elbo) VAE_HIP(hipMemsetAsync(bases[2], 0, (size_t)base_sizes[2] * sizeof(float), c.s));
    for (int i = op_end - 1; i >= op_begin && rc == VAE_OK; --i) {
        const vae_op& op = ops[i];
        if (c.pend && !fused_conv_op(op) && (rc = bn_flush_pending(c)) != VAE_OK) break;
        if (zero_in_elbo && i == op_end - 1) {
            // (the BatchNorm accumulators are cleared on the way: scattered blocks by memsets, behind the forward
            // pass whose consumers were their last readers)
            ZeroRanges zr;
            memset(&zr, 0, sizeof(zr));
            rc = bn_zero_accumulators(c, ops, n_ops, &zr);
            if (rc == VAE_OK) rc = elbo_backward(c, op, bases[2], base_sizes[2], &zr);
            continue;
        }
        if (c.fc && fc_kind(op)) {  // a run of Linear / reparam ops: the input gradients in one launch (fc.h)
            int j = i;
            while (j - 1 >= op_begin && fc_kind(ops[j - 1])) --j;
            const int r = i - j + 1 >= 2 ? fc_backward(c, ops + j, i - j + 1) : 0;
            if (r < 0) return r;
            if (r == 1) {  // (after the launch: a full weight-gradient list flushes at once and reads what it wrote)
                for (int k = i; k >= j && rc == VAE_OK; --k)
                    if (ops[k].kind == VAE_OP_LINEAR) rc = linear_backward(c, ops[k], nullptr, true);
                i = j;
                continue;
            }
        }
        switch (op.kind) {
            case VAE_OP_CONV:
            case VAE_OP_CONVT: rc = (op.flags & VAE_F_FUSED) ? conv_backward_fused(c, op) : conv_backward(c, op); break;
            case VAE_OP_BN: rc = (op.flags & VAE_F_FUSED) ? bn_virtual_backward(c, op) : bn_backward(c, op); break;
            case VAE_OP_LINEAR:
                if (i - 1 >= op_begin && linear_dgrad_siblings(ops[i - 1], op)) {
                    rc = linear_backward(c, op, nullptr, true);                   // weight gradient only
                    if (rc == VAE_OK) rc = linear_backward(c, ops[i - 1], &op);   // + the pair's input gradient
                    --i;
                } else {
                    rc = linear_backward(c, op);
                }
                break;
            case VAE_OP_REPARAM: rc = reparam_backward(c, op); break;
            case VAE_OP_SLICE0: rc = slice0_backward(c, op); break;
            case VAE_OP_ELBO: rc = elbo_backward(c, op); break;
            default: vae_set_error("op %d: unknown kind %d", i, op.kind); rc = VAE_EINVAL;
        }
    }
    if (rc == VAE_OK) rc = bn_flush_pending(c);
    if (rc == VAE_OK) rc = flush_linear_wgrads(c);
    if (rc == VAE_OK) rc = flush_wgrad_finalize(c);
    return rc;
}
extern "C" int vae_run_backward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                                int n_bases, int B, vae_stream_t stream) {
    return vae_run_backward_range(ops, n_ops, bases, base_sizes, n_bases, B, stream, 0, n_ops, 1);
}

// Backward pass + Adam update of the PARAM space (bases[1]) from the GRAD space (bases[2]) as one call: what a train
// step does after the forward (vae_run_backward, then the adam_step_k launch of vae_adam_step_hyper).  One launch that
// both sums a fused 1-D program's gradient rows and applies Adam was measured and dropped: 15 us against 5.5 + 8 us
// for the two launches (the update's loads and stores wait behind the row sum in every one of ~800 workgroups).
extern "C" int vae_run_backward_step(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                                     int n_bases, int B, vae_stream_t stream, float* adam_m, float* adam_v,
                                     const float* hyper, int32_t* tick, float* next_eps, int64_t next_eps_n,
                                     uint64_t* rng) {
    if (!adam_m || !adam_v || !hyper || !tick) VAE_FAIL(VAE_EINVAL, "backward_step: null optimiser state");
    if (next_eps != nullptr && (rng == nullptr || next_eps_n < 1)) VAE_FAIL(VAE_EINVAL, "backward_step: next_eps needs its size and the rng state");
    Ctx c;
    int rc = make_ctx(&c, ops, n_ops, bases, base_sizes, n_bases, B, stream);
    if (rc) return rc;
    if (bases[1] == nullptr || bases[2] == nullptr || base_sizes[1] != base_sizes[2] || base_sizes[1] < 1)
        VAE_FAIL(VAE_EINVAL, "backward_step needs the PARAM (1) and GRAD (2) spaces bound, equally large");
    rc = vae_run_backward_range(ops, n_ops, bases, base_sizes, n_bases, B, stream, 0, n_ops, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(adam_step_k, dim3(ew_grid(base_sizes[1])), dim3(256), 0, c.s, bases[1], (const float*)bases[2], adam_m,
                       adam_v, base_sizes[1], 0.f, 0.f, 0.f, 0.f, tick, hyper, next_eps, next_eps_n, rng);
    VAE_TBYTES(28.0 * (double)base_sizes[1] + 4.0 * (double)(next_eps != nullptr ? next_eps_n : 0));
    VAE_LAUNCH_CHECK("adam_step_k");
    return VAE_OK;
}

// Profiling twin of one whole train step: vae_run_forward (training) + vae_run_backward_step with a HIP event behind
// EVERY kernel launch on the launch stream (VAE_LAUNCH_CHECK marks them while a timer is active on this thread).
// Synchronises the stream; not capturable.
extern "C" int vae_time_step(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes, int n_bases,
                             int B, vae_stream_t stream, float* adam_m, float* adam_v, const float* hyper, int32_t* tick,
                             float* ms, double* bytes, double* flops, char* names, int cap, int* n_launches) {
    if (ms == nullptr || bytes == nullptr || flops == nullptr || names == nullptr || n_launches == nullptr || cap < 1 || cap > 4096)
        VAE_FAIL(VAE_EINVAL, "time_step: bad arguments");
    if (vae_tl_timer != nullptr) VAE_FAIL(VAE_EINVAL, "time_step: a timer is active already");
    VaeTimer tm;
    tm.s = (hipStream_t)stream; tm.n = 0; tm.cap = cap; tm.failed = false;
    tm.ev = new hipEvent_t[cap + 1];
    tm.names = reinterpret_cast<char (*)[VAE_TIMER_NAME]>(names);
    tm.bytes = bytes; tm.flops = flops;
    int made = 0, rc = VAE_OK;
    for (; made <= cap; ++made)
        if (hipEventCreate(&tm.ev[made]) != hipSuccess) break;
    if (made <= cap) rc = VAE_EHIP, vae_set_error("time_step: hipEventCreate failed");
    if (rc == VAE_OK && hipEventRecord(tm.ev[0], tm.s) != hipSuccess) rc = VAE_EHIP, vae_set_error("time_step: hipEventRecord failed");
    if (rc == VAE_OK) {
        vae_tl_timer = &tm;
        rc = vae_run_forward(ops, n_ops, bases, base_sizes, n_bases, B, 1, stream);
        if (rc == VAE_OK)
            rc = hyper != nullptr ? vae_run_backward_step(ops, n_ops, bases, base_sizes, n_bases, B, stream, adam_m, adam_v, hyper,
                                                          tick, nullptr, 0, nullptr)
                                  : vae_run_backward(ops, n_ops, bases, base_sizes, n_bases, B, stream);
        vae_tl_timer = nullptr;
        vae_tl_bytes = vae_tl_flops = 0.0; vae_tl_name[0] = 0;
    }
    if (rc == VAE_OK && tm.failed) rc = VAE_EINVAL, vae_set_error("time_step: more than %d launches, or an event could not be recorded", cap);
    if (rc == VAE_OK && hipEventSynchronize(tm.ev[tm.n]) != hipSuccess) rc = VAE_EHIP, vae_set_error("time_step: hipEventSynchronize failed");
    for (int k = 0; k < tm.n && rc == VAE_OK; ++k)
        if (hipEventElapsedTime(&ms[k], tm.ev[k], tm.ev[k + 1]) != hipSuccess) rc = VAE_EHIP, vae_set_error("hipEventElapsedTime failed");
    if (rc == VAE_OK) *n_launches = tm.n;
    else (void)hipStreamSynchronize(tm.s);  // nothing may still run against events that are about to be destroyed
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(tm.ev[i]);
    delete[] tm.ev;
    return rc;
}

extern "C" int64_t vae_seq_scratch_floats(const vae_op* ops, int n_ops, int B, int64_t nflat) {
    if (ops == nullptr || n_ops < 6 || B < 1 || nflat < 1 || g_bf16 || !sq_enabled()) return 0;
    static thread_local SqProgram sq;
    if (!sq_match(ops, n_ops, &sq) || !sq_plan(sq.d, B)) return 0;
    return sq_scratch_floats(sq.d, nflat);
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
// ---- criterion without a workspace argument and without allocation ---------------------------------
// The partial-sum scratch of vae_elbo_forward lives in a file-scope device array instead of a caller buffer.  The
// array's name is only a host-side stand-in: its device address is looked up once per device with
// hipGetSymbolAddress and that pointer is what the kernels get.
#define VAE_MAX_DEVICES 64
static __device__ __align__(256) float scratch[2 * VAE_ELBO_BLOCKS];  // 8 KiB per GPU, held until the process ends
static float* g_scratch_ptr[VAE_MAX_DEVICES];  // device address of `scratch` on each device (nullptr: not looked up yet)
static std::mutex g_scratch_mu;

// Device pointer of `scratch` on the CURRENT device (looked up on the first call per device).
static int scratch_for_current_device(float** out) {
    int dev = 0;
    VAE_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= VAE_MAX_DEVICES) VAE_FAIL(VAE_EINVAL, "device index %d beyond %d", dev, VAE_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        float* p = nullptr;
        VAE_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (p == nullptr) VAE_FAIL(VAE_EHIP, "hipGetSymbolAddress(scratch) returned a null pointer on device %d", dev);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return VAE_OK;
}

extern "C" int vae_elbo_forward_noalloc(const float* x_rec, const float* target, const float* mu, const float* logvar,
                                        float* out, int B, int n_per_sample, int64_t t_sb, int64_t t_se, int latent,
                                        int bce, vae_stream_t stream) {
    if (!x_rec || !target || !mu || !logvar || !out) VAE_FAIL(VAE_EINVAL, "elbo(noalloc): null pointer");
    if (B < 1 || n_per_sample < 1 || latent < 1 || t_sb < 0 || t_se < 0) VAE_FAIL(VAE_EINVAL, "elbo(noalloc): bad shape");
    if ((int64_t)B * n_per_sample > VAE_ELBO_NOALLOC_MAX_ELEMS || (int64_t)B * latent > VAE_ELBO_NOALLOC_MAX_ELEMS)
        VAE_FAIL(VAE_EINVAL, "elbo(noalloc): %lld x %d reconstruction / %d latent elements exceed the maximum of %lld",
                 (long long)B, n_per_sample, latent, (long long)VAE_ELBO_NOALLOC_MAX_ELEMS);
    float* partials = nullptr;
    const int rc = scratch_for_current_device(&partials);
    if (rc) return rc;
    // every element of the scratch that elbo_finalize_k reads is written by elbo_fwd_k of the same call (all
    // VAE_ELBO_BLOCKS workgroups store their pair), so nothing has to be cleared between calls
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
                       gscale, dx_rec, dmu, dlogvar, (int64_t)B, n_per_sample, t_sb, t_se, latent, bce, (float*)nullptr,
                       (int64_t)0);
    VAE_LAUNCH_CHECK("elbo_bwd_k");
    return VAE_OK;
}

extern "C" int vae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int32_t* tick, vae_stream_t stream) {
    if (!p || !g || !m || !v || !tick || n < 1) VAE_FAIL(VAE_EINVAL, "adam: bad argument");
    // tick block: {int32 t, int32 workgroups done, float bc1, float bc2}; the kernel bumps t itself
    hipLaunchKernelGGL(adam_step_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, tick, (const float*)nullptr);
    VAE_LAUNCH_CHECK("adam_step_k");
    return VAE_OK;
}
extern "C" int vae_adam_step_hyper(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                   int32_t* tick, vae_stream_t stream) {
    if (!p || !g || !m || !v || !tick || !hyper || n < 1) VAE_FAIL(VAE_EINVAL, "adam: bad argument");
    hipLaunchKernelGGL(adam_step_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f, 0.f, 0.f,
                       0.f, tick, hyper);
    VAE_LAUNCH_CHECK("adam_step_k");
    return VAE_OK;
}

extern "C" int vae_randn(float* out, int64_t n, uint64_t* rng, vae_stream_t stream) {
    if (!out || !rng || n < 1) VAE_FAIL(VAE_EINVAL, "randn: bad argument");
    const int64_t nquad = (n + 3) / 4;
    hipLaunchKernelGGL(randn_k, dim3(ew_grid(nquad)), dim3(256), 0, (hipStream_t)stream, out, n, rng);
    VAE_LAUNCH_CHECK("randn_k");
    return VAE_OK;
}

extern "C" int vae_window_gather(const float* series, int C, int64_t T, int L, const int64_t* starts, int B, float* out,
                                 vae_stream_t stream) {
    if (!series || !starts || !out || C < 1 || L < 1 || T < L || B < 1 || (int64_t)C * L >= ((int64_t)1 << 31))
        VAE_FAIL(VAE_EINVAL, "window_gather: bad argument (C=%d T=%lld L=%d B=%d)", C, (long long)T, L, B);
    hipLaunchKernelGGL(window_gather_k, dim3(ew_grid((int64_t)B * C * L)), dim3(256), 0, (hipStream_t)stream, series,
                       starts, out, C, T, L, B);
    VAE_LAUNCH_CHECK("window_gather_k");
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
