// seq1d_host.h - host side of seq1d.h: recognises a whole 1-D VAE program (TCVAE / multiscale_vae lowered by
// vae_amd/program.py) in an op array, plans chunking and LDS, and enqueues one launch per BatchNorm-delimited phase.
// Included by vae_amd.hip after its Ctx / resolve helpers.
//
// Program shape recognised (what HipVAE.emit_program and FusedTrainStep emit for the 1-D families):
//   nenc x { NE x [CONV 1xk stride 2 + LeakyReLU, BN], LINEAR mu, LINEAR logvar }   (Encoders.py:76-87, 100-107)
//   REPARAM, LINEAR,  ND x [CONVT 1x2 stride 2 + LeakyReLU, (BN)],  SLICE0,  [ELBO]  (vae.py:20-23, Decoders.py:57-65)
// The first op's r[11] (otherwise unused) carries the WORK reference of the scratch region
// (vae_seq_scratch_floats); a program without it runs op by op.
#pragma once
#include "seq1d.h"

#define SQ_LDS_BUDGET 40704  // floats (159 KiB of the 160 KiB) of dynamic LDS a phase may use

struct SqProgram {
    SqDesc d;
    int enc_conv[SQ_MAXENC][SQ_MAXN], enc_bn[SQ_MAXENC][SQ_MAXN], head_mu[SQ_MAXENC], head_lv[SQ_MAXENC];
    int reparam, lin, dec_conv[SQ_MAXN], dec_bn[SQ_MAXN], slice0, elbo;
};

static bool sq_conv1d_ok(const vae_op& op, int transposed) {
    const int* g = op.d;  // Cin H W Cout OH OW KH KW sh sw ph pw
    if (g[1] != 1 || g[4] != 1 || g[6] != 1 || g[8] != 1 || g[9] != 2 || g[10] != 0 || g[11] != 0) return false;
    if (g[0] < 1 || g[0] > SQ_MAXC || g[3] < 1 || g[3] > SQ_MAXC || g[2] < 1 || g[5] < 1 || g[7] < 1) return false;
    if (op.act != VAE_ACT_LEAKY && op.act != VAE_ACT_NONE) return false;
    if (op.flags & VAE_F_FUSED) return false;
    if (transposed) return g[7] == 2 && g[5] == 2 * g[2];
    return g[2] >= g[7] && g[5] == (g[2] - g[7]) / 2 + 1;
}
static void sq_layer_shape(SqLayer& y, const vae_op& op) {
    y.cin = op.d[0]; y.lin = op.d[2]; y.cout = op.d[3]; y.lout = op.d[5]; y.k = op.d[7];
    y.has_bn = 0; y.bn_eps = 1e-5f; y.bn_mom = 0.1f;
}
static float sq_slope(const vae_op& op) { return op.act == VAE_ACT_LEAKY ? op.f[0] : 1.f; }  // no activation = slope 1

// Structure only (no pointers): fills the shapes of p->d and the op indices.  Returns false when the op array is not
// a whole 1-D VAE program of supported shape.
static bool sq_match(const vae_op* ops, int n_ops, SqProgram* p) {
    memset(p, 0, sizeof(*p));
    SqDesc& d = p->d;
    p->elbo = -1;
    int i = 0;
    bool have_slope = false;
    auto slope_ok = [&](const vae_op& op) {
        const float s = sq_slope(op);
        if (!have_slope) { d.slope = s; have_slope = true; }
        return s == d.slope;
    };
    while (i < n_ops && ops[i].kind == VAE_OP_CONV) {
        if (d.nenc == SQ_MAXENC) return false;
        const int e = d.nenc;
        int nl = 0;
        while (i < n_ops && ops[i].kind == VAE_OP_CONV) {
            if (nl == SQ_MAXN || i + 1 >= n_ops || ops[i + 1].kind != VAE_OP_BN) return false;
            const vae_op &cv = ops[i], &bn = ops[i + 1];
            if (!sq_conv1d_ok(cv, 0) || !slope_ok(cv) || (nl > 0 && cv.d[7] != 2)) return false;
            if (bn.r[0] != cv.r[1] || bn.d[0] != cv.d[3] || bn.d[1] != cv.d[5] || (bn.flags & (VAE_F_BN_RELU | VAE_F_FUSED)))
                return false;
            if (nl == 0 ? cv.r[0] != ops[0].r[0] : cv.r[0] != ops[i - 1].r[1]) return false;
            SqLayer& y = d.enc[e][nl];
            sq_layer_shape(y, cv);
            if (nl > 0 && (y.cin != d.enc[e][nl - 1].cout || y.lin != d.enc[e][nl - 1].lout)) return false;
            y.has_bn = 1; y.bn_eps = bn.f[0]; y.bn_mom = bn.f[1];
            p->enc_conv[e][nl] = i; p->enc_bn[e][nl] = i + 1;
            i += 2; ++nl;
        }
        if (e == 0) d.NE = nl;
        else if (nl != d.NE) return false;
        if (i + 1 >= n_ops || ops[i].kind != VAE_OP_LINEAR || ops[i + 1].kind != VAE_OP_LINEAR) return false;
        const SqLayer& last = d.enc[e][nl - 1];
        const int K = last.cout * last.lout;
        for (int h = 0; h < 2; ++h) {
            const vae_op& ln = ops[i + h];
            if (ln.r[0] != ops[i - 1].r[1] || ln.d[0] != K || ln.d[2] != K || ln.act != VAE_ACT_NONE || ln.r[5] < 0) return false;
            if (e == 0 && h == 0) d.lat = ln.d[1];
            if (ln.d[1] != d.lat) return false;
        }
        d.head[e].K = K;
        p->head_mu[e] = i; p->head_lv[e] = i + 1;
        i += 2;
        ++d.nenc;
    }
    if (d.nenc < 1 || i >= n_ops || ops[i].kind != VAE_OP_REPARAM) return false;
    const vae_op& rp = ops[i];
    p->reparam = i++;
    d.lat_total = rp.d[0];
    if (d.lat_total != d.nenc * d.lat || d.lat_total > 256) return false;
    for (int e = 0; e < d.nenc; ++e) {  // heads write the columns [e*lat, (e+1)*lat) of mu / logvar
        const vae_op &hm = ops[p->head_mu[e]], &hl = ops[p->head_lv[e]];
        if (hm.r[1] != rp.r[0] + e * d.lat || hl.r[1] != rp.r[1] + e * d.lat || hm.d[3] != d.lat_total || hl.d[3] != d.lat_total)
            return false;
    }
    d.kl_grad = (rp.flags & VAE_F_KL_GRAD) ? 1 : 0;
    if (i >= n_ops || ops[i].kind != VAE_OP_LINEAR) return false;
    const vae_op& dl = ops[i];
    if (dl.r[0] != rp.r[3] || dl.d[0] != d.lat_total || dl.d[2] != d.lat_total || dl.act != VAE_ACT_NONE || dl.r[5] < 0 ||
        dl.d[3] != dl.d[1])
        return false;
    p->lin = i++;
    while (i < n_ops && ops[i].kind == VAE_OP_CONVT) {
        if (d.ND == SQ_MAXN) return false;
        const vae_op& cv = ops[i];
        if (!sq_conv1d_ok(cv, 1) || !slope_ok(cv) || cv.r[0] != ops[i - 1].r[1]) return false;
        SqLayer& y = d.dec[d.ND];
        sq_layer_shape(y, cv);
        if (d.ND == 0 ? (y.lin != 1 || y.cin != dl.d[1]) : (y.cin != d.dec[d.ND - 1].cout || y.lin != d.dec[d.ND - 1].lout))
            return false;
        p->dec_conv[d.ND] = i; p->dec_bn[d.ND] = -1;
        ++i;
        if (i < n_ops && ops[i].kind == VAE_OP_BN) {
            const vae_op& bn = ops[i];
            if (bn.r[0] != cv.r[1] || bn.d[0] != cv.d[3] || bn.d[1] != cv.d[5] || (bn.flags & (VAE_F_BN_RELU | VAE_F_FUSED)))
                return false;
            y.has_bn = 1; y.bn_eps = bn.f[0]; y.bn_mom = bn.f[1];
            p->dec_bn[d.ND] = i++;
        }
        ++d.ND;
    }
    if (d.ND < 1 || d.dec[d.ND - 1].has_bn) return false;
    for (int j = 0; j + 1 < d.ND; ++j)
        if (!d.dec[j].has_bn) return false;
    if (i >= n_ops || ops[i].kind != VAE_OP_SLICE0) return false;
    const vae_op& sl = ops[i];
    if (sl.r[0] != ops[i - 1].r[1] || sl.d[0] != d.dec[d.ND - 1].cout || sl.d[1] != d.dec[d.ND - 1].lout) return false;
    p->slice0 = i++;
    if (i < n_ops && ops[i].kind == VAE_OP_ELBO) {
        if (ops[i].r[0] != sl.r[1]) return false;
        p->elbo = i++;
    }
    if (i != n_ops) return false;
    d.C = d.enc[0][0].cin; d.L = d.enc[0][0].lin;
    for (int e = 1; e < d.nenc; ++e)
        if (d.enc[e][0].cin != d.C || d.enc[e][0].lin != d.L) return false;
    return true;
}

static int sq_max_lds(const SqDesc& d, int SB) {
    int m = sq_lds_fh(d, SB);
    auto up = [&](int v) { if (v > m) m = v; };
    up(sq_lds_bh(d, SB));
    for (int i = 0; i < d.NE; ++i) { up(sq_lds_fe(d, i, SB)); up(sq_lds_be(d, i, SB)); }
    for (int j = 1; j < d.ND; ++j) { up(sq_lds_fd(d, j, SB)); up(sq_lds_bd(d, j, SB)); }
    return m;
}
static int sq_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v != nullptr && *v) ? atoi(v) : dflt;
}
// Chunk size and grid for batch B: about one chunk per workgroup of a grid that fills the GPU, bounded by what the
// largest phase fits in LDS.  false: even one sample per chunk does not fit (the program then runs op by op).
static bool sq_plan(SqDesc& d, int B) {
    if (sq_max_lds(d, 1) > SQ_LDS_BUDGET) return false;
    int fit = 1;
    while (fit < 64 && sq_max_lds(d, fit + 1) <= SQ_LDS_BUDGET) ++fit;
    int target = sq_env_int("VAE_SEQ1D_GRID", SQ_MAX_GRID);
    target = target < 1 ? 1 : (target > SQ_MAX_GRID ? SQ_MAX_GRID : target);
    int sb = (int)ceil_div64(B, target);
    if (sb > fit) {  // several rounds per workgroup: the smallest chunk that needs no more rounds than the largest
        const int64_t rounds = ceil_div64(ceil_div64(B, fit), target);
        sb = (int)ceil_div64(B, (int64_t)target * rounds);
    }
    const int forced = sq_env_int("VAE_SEQ1D_SB", 0);
    if (forced > 0) sb = forced;
    if (sb > fit) sb = fit;
    if (sb < 1) sb = 1;
    d.B = B; d.SB = sb;
    d.nchunks = (int)ceil_div64(B, sb);
    d.grid = d.nchunks < target ? d.nchunks : target;
    if (sq_env_int("VAE_SEQ1D_VERBOSE", 0))
        fprintf(stderr, "[seq1d] B=%d SB=%d (fit %d) chunks=%d grid=%d lds(max)=%d floats\n", B, d.SB, fit, d.nchunks, d.grid,
                sq_max_lds(d, d.SB));
    if (sq_env_int("VAE_SEQ1D_VERBOSE", 0) > 1) {
        fprintf(stderr, "[seq1d] lds floats at SB=%d: fh %d bh %d", fit + 1, sq_lds_fh(d, fit + 1), sq_lds_bh(d, fit + 1));
        for (int i = 0; i < d.NE; ++i) fprintf(stderr, " fe%d %d be%d %d", i, sq_lds_fe(d, i, fit + 1), i, sq_lds_be(d, i, fit + 1));
        for (int j = 1; j < d.ND; ++j) fprintf(stderr, " fd%d %d bd%d %d", j, sq_lds_fd(d, j, fit + 1), j, sq_lds_bd(d, j, fit + 1));
        fprintf(stderr, " (budget %d)\n", SQ_LDS_BUDGET);
    }
    return true;
}
static int64_t sq_scratch_floats(const SqDesc& d, int64_t nflat) {
    int64_t n = 0;
    for (int e = 0; e < d.nenc; ++e)
        for (int i = 0; i < d.NE; ++i) n += (int64_t)5 * d.enc[e][i].cout * d.grid;
    for (int j = 0; j < d.ND; ++j) n += (int64_t)5 * d.dec[j].cout * d.grid;
    // + the ranks' gathered statistics rows of a data-parallel step (sized for any world: the collective may be
    // installed after the program was built)
    int64_t g = 0;
    for (int e = 0; e < d.nenc; ++e)
        for (int i = 0; i < d.NE; ++i) g += 5 * d.enc[e][i].cout + 4;
    for (int j = 0; j < d.ND; ++j) g += 5 * d.dec[j].cout + 4;
    return n + (int64_t)d.grid * nflat + 64 + g * VAE_MAX_WORLD;
}

static bool sq_enabled() { return sq_env_int("VAE_AMD_SEQ1D", 1) != 0; }

// Binds every pointer of the descriptor from the program's references (bounds-checked like the per-op path).
static int sq_bind(const Ctx& c, const vae_op* ops, SqProgram* p, int training, bool backward) {
    SqDesc& d = p->d;
    const int B = c.B;
    bool ok = true;
    const int64_t nflat = c.sizes[1];
    d.training = training;
    d.nflat = nflat;
    float* scratch = resolve(c, ops[0].r[11], sq_scratch_floats(d, nflat), "seq1d.scratch", &ok);
    if (!ok) return VAE_EINVAL;
    float* sp = scratch;
    auto grad_off = [&](int64_t ref, int64_t count, const char* what) -> int64_t {
        const int64_t off = ref_off(ref);
        if (ref < 0 || ref_space(ref) != 1 || off + count > nflat || (off & 3)) {
            vae_set_error("seq1d: %s is not a 4-float aligned PARAM reference", what);
            ok = false;
            return 0;
        }
        return off;
    };
    auto bind_layer = [&](SqLayer& y, const vae_op& cv, const vae_op* bn) {
        const int64_t nw = (int64_t)y.cin * y.cout * y.k, nout = (int64_t)B * y.cout * y.lout;
        y.w = resolve(c, cv.r[4], nw, "seq1d.conv.w", &ok);
        y.b = resolve(c, cv.r[5], y.cout, "seq1d.conv.b", &ok);
        y.w_off = grad_off(cv.r[4], nw, "conv.w");
        y.b_off = grad_off(cv.r[5], y.cout, "conv.b");
        y.y = resolve(c, cv.r[1], nout, "seq1d.conv.out", &ok);
        y.g = nullptr;
        if (bn != nullptr) {
            y.gamma = resolve(c, bn->r[4], y.cout, "seq1d.bn.gamma", &ok);
            y.beta = resolve(c, bn->r[5], y.cout, "seq1d.bn.beta", &ok);
            y.ga_off = grad_off(bn->r[4], y.cout, "bn.gamma");
            y.be_off = grad_off(bn->r[5], y.cout, "bn.beta");
            y.rmean = resolve(c, bn->r[7], y.cout, "seq1d.bn.running_mean", &ok);
            y.rvar = resolve(c, bn->r[8], y.cout, "seq1d.bn.running_var", &ok);
            y.save = resolve(c, bn->r[6], (int64_t)8 * y.cout, "seq1d.bn.save", &ok);
            if (backward) y.g = resolve(c, bn->r[3], nout, "seq1d.bn.dout", &ok);
        } else if (backward && cv.r[3] >= 0) {
            y.g = resolve(c, cv.r[3], nout, "seq1d.conv.dout", &ok);
        }
        y.fstat = sp; sp += (int64_t)3 * y.cout * d.grid;
        y.bstat = sp; sp += (int64_t)2 * y.cout * d.grid;
    };
    for (int e = 0; e < d.nenc; ++e) {
        for (int i = 0; i < d.NE; ++i) bind_layer(d.enc[e][i], ops[p->enc_conv[e][i]], &ops[p->enc_bn[e][i]]);
        SqHead& h = d.head[e];
        const vae_op &hm = ops[p->head_mu[e]], &hl = ops[p->head_lv[e]];
        h.wmu = resolve(c, hm.r[4], (int64_t)d.lat * h.K, "seq1d.mu.w", &ok);
        h.bmu = resolve(c, hm.r[5], d.lat, "seq1d.mu.b", &ok);
        h.wlv = resolve(c, hl.r[4], (int64_t)d.lat * h.K, "seq1d.logvar.w", &ok);
        h.blv = resolve(c, hl.r[5], d.lat, "seq1d.logvar.b", &ok);
        h.wmu_off = grad_off(hm.r[4], (int64_t)d.lat * h.K, "mu.w");
        h.bmu_off = grad_off(hm.r[5], d.lat, "mu.b");
        h.wlv_off = grad_off(hl.r[4], (int64_t)d.lat * h.K, "logvar.w");
        h.blv_off = grad_off(hl.r[5], d.lat, "logvar.b");
    }
    for (int j = 0; j < d.ND; ++j) bind_layer(d.dec[j], ops[p->dec_conv[j]], p->dec_bn[j] >= 0 ? &ops[p->dec_bn[j]] : nullptr);
    d.part = sp;
    {   // data parallel: per BatchNorm boundary (layer i of every encoder; a decoder layer) [world][floats of the boundary]
        d.world = training ? g_world : 1;
        d.rank = training ? g_rank : 0;
        d.gath = sp + (int64_t)d.grid * nflat + 64;
        int64_t off = 0;
        auto boundary = [&](SqLayer** ys, int n) {
            int fs = 0, bs = 0;
            for (int k = 0; k < n; ++k) { fs += 3 * ys[k]->cout; bs += 2 * ys[k]->cout + 4; }
            int fo = 0, bo = 0;
            for (int k = 0; k < n; ++k) {
                SqLayer& y = *ys[k];
                y.fg_stride = fs; y.bg_stride = bs;
                y.fg_off = (int)(off + fo);
                y.bg_off = (int)(off + (int64_t)fs * VAE_MAX_WORLD + bo);
                fo += 3 * y.cout; bo += 2 * y.cout + 4;
            }
            off += (int64_t)(fs + bs) * VAE_MAX_WORLD;
        };
        for (int i = 0; i < d.NE; ++i) {
            SqLayer* ys[SQ_MAXENC];
            for (int e = 0; e < d.nenc; ++e) ys[e] = &d.enc[e][i];
            boundary(ys, d.nenc);
        }
        for (int j = 0; j < d.ND; ++j) { SqLayer* ys[1] = {&d.dec[j]}; boundary(ys, 1); }
    }
    const vae_op &rp = ops[p->reparam], &dl = ops[p->lin], &sl = ops[p->slice0];
    const int c0 = d.dec[0].cin;
    const int64_t nlat = (int64_t)B * d.lat_total;
    d.wl = resolve(c, dl.r[4], (int64_t)c0 * d.lat_total, "seq1d.lin.w", &ok);
    d.bl = resolve(c, dl.r[5], c0, "seq1d.lin.b", &ok);
    d.wl_off = grad_off(dl.r[4], (int64_t)c0 * d.lat_total, "lin.w");
    d.bl_off = grad_off(dl.r[5], c0, "lin.b");
    d.x = resolve(c, ops[0].r[0], (int64_t)B * d.C * d.L, "seq1d.x", &ok);
    d.mu = resolve(c, rp.r[0], nlat, "seq1d.mu", &ok);
    d.lv = resolve(c, rp.r[1], nlat, "seq1d.logvar", &ok);
    d.eps = resolve(c, rp.r[2], nlat, "seq1d.eps", &ok);
    d.z = resolve(c, rp.r[3], nlat, "seq1d.z", &ok);
    d.h0 = resolve(c, dl.r[1], (int64_t)B * c0, "seq1d.h0", &ok);
    d.xrec = resolve(c, sl.r[1], (int64_t)B * d.dec[d.ND - 1].cout, "seq1d.x_rec", &ok);
    d.dxrec = nullptr; d.gmu = d.glv = nullptr; d.gscale = nullptr;
    d.has_elbo = 0; d.bce = 0; d.t_sb = d.t_se = 0; d.target = nullptr; d.egscale = nullptr; d.dxrec_w = nullptr; d.epart = nullptr;
    d.eout = nullptr; d.efin = 0; d.eacc = 0;
    if (p->elbo >= 0) {  // the criterion runs inside the last forward phase
        const vae_op& eo = ops[p->elbo];
        const int nps = d.dec[d.ND - 1].cout;
        if (eo.d[0] != nps || eo.d[3] != d.lat_total || eo.d[1] < 0 || eo.d[2] < 0) {
            vae_set_error("seq1d: ELBO op shape %d/%d does not match the model (%d, %d)", eo.d[0], eo.d[3], nps, d.lat_total);
            return VAE_EINVAL;
        }
        d.has_elbo = 1; d.bce = (eo.flags & VAE_F_BCE) ? 1 : 0; d.t_sb = eo.d[1]; d.t_se = eo.d[2];
        d.target = resolve(c, eo.r[1], (int64_t)(B - 1) * eo.d[1] + (int64_t)(nps - 1) * eo.d[2] + 1, "seq1d.elbo.target", &ok);
        d.egscale = resolve_opt(c, eo.r[7], 1, "seq1d.elbo.gscale", &ok);
        d.dxrec_w = resolve(c, eo.r[5], (int64_t)B * nps, "seq1d.elbo.dx_rec", &ok);
        d.epart = resolve(c, eo.r[6], 2 * VAE_ELBO_BLOCKS, "seq1d.elbo.partials", &ok);
        if (backward && (eo.flags & VAE_F_LOSS_IN_BWD)) {  // the sum of the partials rides on the first backward launch
            d.eout = resolve(c, eo.r[4], 4, "seq1d.elbo.out", &ok);
            d.efin = 1; d.eacc = eo.d[5];
        }
    }
    // eval forward without a criterion: the decoder is evaluated at time index 0 only, inside the heads phase (f3)
    d.prune = (!training && !backward && !d.has_elbo && sq_env_int("VAE_SEQ1D_PRUNE", 1) != 0) ? 1 : 0;
    if (backward) {
        d.dxrec = resolve(c, sl.r[3], (int64_t)B * d.dec[d.ND - 1].cout, "seq1d.dx_rec", &ok);
        if (rp.flags & VAE_F_EXT_GRAD) {
            d.gmu = resolve(c, rp.r[7], nlat, "seq1d.g_mu", &ok);
            d.glv = resolve(c, rp.r[8], nlat, "seq1d.g_logvar", &ok);
        }
        if (rp.flags & VAE_F_KL_GRAD) d.gscale = resolve_opt(c, rp.r[9], 1, "seq1d.gscale", &ok);
    }
    return ok ? VAE_OK : VAE_EINVAL;
}

// dynamic LDS beyond 64 KiB has to be granted per kernel, once per device
static int sq_grant_lds() {
    static bool done[64];
    static std::mutex mu;   // two host threads may run programs on two devices at once
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    VAE_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) VAE_FAIL(VAE_EINVAL, "device index %d out of range", dev);
    if (done[dev]) return VAE_OK;
    const int bytes = 160 * 1024;
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_fe_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_fe_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_fh_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_fh_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_fd_k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_bd_k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_bh_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_bh_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_be_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    VAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sq_be_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[dev] = true;
    return VAE_OK;
}

// true: the op array is a seq1d program (matched, scratch attached)
static bool sq_use(const vae_op* ops, int n_ops, int B, SqProgram* p) {
    // (fp32 only: under the bf16 compute dtype the program runs op by op on the kernels that have bf16 instances)
    if (n_ops < 6 || ops[0].r[11] < 0 || g_bf16 || !sq_enabled()) return false;
    return sq_match(ops, n_ops, p) && sq_plan(p->d, B);
}

// Diagnostic timing of the phases (vae_time_seq): an event before every launch on the launch stream, with the
// algorithmic bytes of the launch (each input read once, each output written once; weights included).
#define SQ_MAX_MARKS 40
struct SqTimer {
    hipEvent_t ev[SQ_MAX_MARKS + 1];
    double bytes[SQ_MAX_MARKS];
    char tag[SQ_MAX_MARKS][8];
    int n;
    bool failed;
};
static thread_local int sq_tl_nenc = 1;  // (for the kernel names of vae_time_step)
static void sq_mark(SqTimer* tm, hipStream_t s, const char* tag, double floats) {
    if (vae_tl_timer != nullptr) {  // vae_time_step: the launch that follows carries these bytes and the phase tag
        VAE_TBYTES(4.0 * floats);
        const char* kn = tag[0] == 'r' ? "sq_reduce_k" : tag[0] == 'e' ? "elbo_finalize_k" : tag[0] == 'x' ? "sq_rank_rows_k" :
                         tag[1] == 'e' ? (tag[0] == 'f' ? "sq_fe_k" : "sq_be_k") : tag[1] == 'h' ? (tag[0] == 'f' ? "sq_fh_k" : "sq_bh_k") :
                                         (tag[0] == 'f' ? "sq_fd_k" : "sq_bd_k");
        if (tag[0] != 'r' && tag[0] != 'e' && tag[0] != 'x' && tag[1] != 'd') VAE_TNAME("%s<%d> (phase %s)", kn, sq_tl_nenc, tag);
        else VAE_TNAME("%s (phase %s)", kn, tag);
    }
    if (tm == nullptr || tm->n >= SQ_MAX_MARKS) return;
    if (hipEventRecord(tm->ev[tm->n], s) != hipSuccess) tm->failed = true;
    tm->bytes[tm->n] = 4.0 * floats;
    snprintf(tm->tag[tm->n], sizeof(tm->tag[0]), "%s", tag);
    ++tm->n;
}
static double sq_layer_w(const SqLayer& y) { return (double)y.cin * y.cout * y.k + 3.0 * y.cout; }

// Data parallel: the statistics rows that the phase just enqueued leaves for n layers (one BatchNorm boundary) become
// one row per rank (sq_rank_rows_k) and are all-gathered in place through the collective hook; the next phase's
// prologue merges the `world` rows.  backward: the rows of the backward sums.
static int sq_exchange(const Ctx& c, const SqDesc& d, const SqLayer* const* ys, int n, bool backward, SqTimer* tm, const char* tag) {
    if (d.world <= 1) return VAE_OK;
    if (g_allgather == nullptr) VAE_FAIL(VAE_EINVAL, "seq1d: world %d without a collective hook", d.world);
    SqRankRows a;
    memset(&a, 0, sizeof(a));
    int Cmax = 0;
    for (int k = 0; k < n; ++k) {
        const SqLayer& y = *ys[k];
        const int slot = k == 0 ? 0 : SQ_MAXENC - 1;
        a.stat[slot] = backward ? y.bstat : y.fstat;
        a.out[slot] = d.gath + (backward ? y.bg_off + (int64_t)d.rank * y.bg_stride : y.fg_off + (int64_t)d.rank * y.fg_stride);
        a.C[slot] = y.cout;
        a.n_local[slot] = (float)d.B * (float)y.lout;
        Cmax = y.cout > Cmax ? y.cout : Cmax;
    }
    a.rows = d.grid; a.nq = backward ? 2 : 3;
    const SqLayer& y0 = *ys[0];
    const int64_t stride = backward ? y0.bg_stride : y0.fg_stride;
    sq_mark(tm, c.s, tag, (double)(a.nq * d.grid + a.nq) * Cmax * n);
    hipLaunchKernelGGL(sq_rank_rows_k, dim3((Cmax + 3) / 4, n), dim3(256), 0, c.s, a);
    VAE_LAUNCH_CHECK("sq_rank_rows_k");
    if (g_allgather(g_allgather_user, d.gath + (backward ? y0.bg_off : y0.fg_off), stride, (vae_stream_t)c.s) != 0)
        VAE_FAIL(VAE_EINVAL, "seq1d: the all-gather hook failed");
    return VAE_OK;
}
static int sq_exchange_level(const Ctx& c, const SqDesc& d, int i, bool backward, SqTimer* tm) {
    const SqLayer* ys[SQ_MAXENC];
    for (int e = 0; e < d.nenc; ++e) ys[e] = &d.enc[e][i];
    return sq_exchange(c, d, ys, d.nenc, backward, tm, "xe");
}
static int sq_exchange_dec(const Ctx& c, const SqDesc& d, int j, bool backward, SqTimer* tm) {
    const SqLayer* ys[1] = {&d.dec[j]};
    return sq_exchange(c, d, ys, 1, backward, tm, "xd");
}

static int sq_forward(const Ctx& c, const vae_op* ops, SqProgram* p, int training, SqTimer* tm = nullptr) {
    int rc = sq_bind(c, ops, p, training, false);
    if (rc) return rc;
    if ((rc = sq_grant_lds()) != VAE_OK) return rc;
    const SqDesc& d = p->d;
    sq_tl_nenc = d.nenc;
    if (training && (int64_t)c.B * d.dec[0].lout < 2) VAE_FAIL(VAE_EINVAL, "bn: training needs more than 1 value per channel");
    const double Bd = (double)c.B;
    char tag[8];
    for (int i = 0; i < d.NE; ++i) {
        double fl = 0;
        for (int e = 0; e < d.nenc; ++e) {
            const SqLayer& y = d.enc[e][i];
            fl += Bd * ((double)y.cin * y.lin + (double)y.cout * y.lout) + sq_layer_w(y);
        }
        snprintf(tag, sizeof(tag), "fe%d", i);
        sq_mark(tm, c.s, tag, fl);
        if (d.nenc == 1) hipLaunchKernelGGL(sq_fe_k<1>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_fe(d, i, d.SB) * sizeof(float), c.s, d, i);
        else hipLaunchKernelGGL(sq_fe_k<2>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_fe(d, i, d.SB) * sizeof(float), c.s, d, i);
        VAE_LAUNCH_CHECK("sq_fe_k");
        if (training && (rc = sq_exchange_level(c, d, i, false, tm)) != VAE_OK) return rc;
    }
    {
        double fl = Bd * (4.0 * d.lat_total + d.dec[0].cin + (double)d.dec[0].cout * d.dec[0].lout) +
                    (double)d.dec[0].cin * (d.lat_total + 1) + sq_layer_w(d.dec[0]);
        for (int e = 0; e < d.nenc; ++e) fl += (Bd + 2.0 * d.lat) * d.head[e].K;
        sq_mark(tm, c.s, "fh", fl);
    }
    if (d.nenc == 1) hipLaunchKernelGGL(sq_fh_k<1>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_fh(d, d.SB) * sizeof(float), c.s, d);
    else hipLaunchKernelGGL(sq_fh_k<2>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_fh(d, d.SB) * sizeof(float), c.s, d);
    VAE_LAUNCH_CHECK("sq_fh_k");
    if (training && d.dec[0].has_bn && (rc = sq_exchange_dec(c, d, 0, false, tm)) != VAE_OK) return rc;
    for (int j = 1; j < d.ND && !d.prune; ++j) {
        const SqLayer& y = d.dec[j];
        double fl = Bd * ((double)y.cin * y.lin + (double)y.cout * y.lout) + sq_layer_w(y);
        if (j == d.ND - 1) fl = Bd * ((double)y.cin * y.lin + (d.has_elbo ? 3.0 : 1.0) * y.cout) + sq_layer_w(y);  // only x[:, :, 0] leaves
        snprintf(tag, sizeof(tag), "fd%d", j);
        sq_mark(tm, c.s, tag, fl);
        hipLaunchKernelGGL(sq_fd_k, dim3(d.grid), dim3(SQ_THREADS), sq_lds_fd(d, j, d.SB) * sizeof(float), c.s, d, j);
        VAE_LAUNCH_CHECK("sq_fd_k");
        if (training && y.has_bn && (rc = sq_exchange_dec(c, d, j, false, tm)) != VAE_OK) return rc;
    }
    if (p->elbo >= 0 && !(ops[p->elbo].flags & VAE_F_LOSS_IN_BWD)) {  // fixed-order sum of the workgroups' {recon, kld} partials -> out[0..3]
        const vae_op& eo = ops[p->elbo];
        bool ok = true;
        float* out = resolve(c, eo.r[4], 4, "seq1d.elbo.out", &ok);
        if (!ok) return VAE_EINVAL;
        sq_mark(tm, c.s, "elbofin", 2.0 * d.grid + 4);
        hipLaunchKernelGGL(elbo_finalize_k, dim3(1), dim3(256), 0, c.s, d.epart, out, d.grid, eo.d[5]);
        VAE_LAUNCH_CHECK("elbo_finalize_k");
    }
    return rc;
}

static int sq_backward(const Ctx& c, const vae_op* ops, SqProgram* p, SqTimer* tm = nullptr) {
    int rc = sq_bind(c, ops, p, 1, true);
    if (rc) return rc;
    if ((rc = sq_grant_lds()) != VAE_OK) return rc;
    const SqDesc& d = p->d;
    sq_tl_nenc = d.nenc;
    bool ok = true;
    float* grad = resolve(c, (int64_t)2 << 56, d.nflat, "seq1d.grad", &ok);
    if (!ok) return VAE_EINVAL;
    if (c.sizes[2] != d.nflat) VAE_FAIL(VAE_EINVAL, "seq1d: GRAD space (%lld floats) differs from PARAM (%lld)", (long long)c.sizes[2], (long long)d.nflat);
    // (with the criterion in the program, d(loss)/d(x_rec) was written by the last forward phase)
    const double Bd = (double)c.B;
    char tag[8];
    for (int j = d.ND - 1; j >= 1; --j) {
        const SqLayer& y = d.dec[j];
        const double gy = j == d.ND - 1 ? (double)y.cout : 2.0 * y.cout * y.lout;  // last layer: d(x_rec) only, no BatchNorm
        snprintf(tag, sizeof(tag), "bd%d", j);
        sq_mark(tm, c.s, tag, Bd * (gy + 2.0 * y.cin * y.lin) + 2.0 * sq_layer_w(y));
        hipLaunchKernelGGL(sq_bd_k, dim3(d.grid), dim3(SQ_THREADS), sq_lds_bd(d, j, d.SB) * sizeof(float), c.s, d, j);
        VAE_LAUNCH_CHECK("sq_bd_k");
        if ((rc = sq_exchange_dec(c, d, j - 1, true, tm)) != VAE_OK) return rc;   // the sums of dec[j-1]'s BatchNorm
    }
    {
        double fl = Bd * (2.0 * d.dec[0].cout * d.dec[0].lout + d.dec[0].cin + 4.0 * d.lat_total) +
                    2.0 * ((double)d.dec[0].cin * (d.lat_total + 1) + sq_layer_w(d.dec[0]));
        for (int e = 0; e < d.nenc; ++e) fl += (2.0 * Bd + 4.0 * d.lat) * d.head[e].K;
        sq_mark(tm, c.s, "bh", fl);
    }
    if (d.nenc == 1) hipLaunchKernelGGL(sq_bh_k<1>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_bh(d, d.SB) * sizeof(float), c.s, d);
    else hipLaunchKernelGGL(sq_bh_k<2>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_bh(d, d.SB) * sizeof(float), c.s, d);
    VAE_LAUNCH_CHECK("sq_bh_k");
    if ((rc = sq_exchange_level(c, d, d.NE - 1, true, tm)) != VAE_OK) return rc;
    for (int i = d.NE - 1; i >= 0; --i) {
        double fl = 0;
        for (int e = 0; e < d.nenc; ++e) {
            const SqLayer& y = d.enc[e][i];
            fl += Bd * (2.0 * y.cout * y.lout + (i > 0 ? 2.0 : 1.0) * y.cin * y.lin) + 2.0 * sq_layer_w(y);
        }
        snprintf(tag, sizeof(tag), "be%d", i);
        sq_mark(tm, c.s, tag, fl);
        if (d.nenc == 1) hipLaunchKernelGGL(sq_be_k<1>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_be(d, i, d.SB) * sizeof(float), c.s, d, i);
        else hipLaunchKernelGGL(sq_be_k<2>, dim3(d.grid), dim3(SQ_THREADS), sq_lds_be(d, i, d.SB) * sizeof(float), c.s, d, i);
        VAE_LAUNCH_CHECK("sq_be_k");
        if (i > 0 && (rc = sq_exchange_level(c, d, i - 1, true, tm)) != VAE_OK) return rc;
    }
    sq_mark(tm, c.s, "reduce", (double)d.nflat * (d.grid + 1));
    hipLaunchKernelGGL(sq_reduce_k, dim3((unsigned)ceil_div64(d.nflat, SQR_COLS)), dim3(256), 0, c.s, d.part, grad, d.nflat, d.grid);
    VAE_LAUNCH_CHECK("sq_reduce_k");
    return VAE_OK;
}
