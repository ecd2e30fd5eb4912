// elem.h - fused elementwise / loss / optimiser kernels (HBM-bound, coalesced, grid-stride).
#pragma once
#include "common.h"

// z = mu + eps * exp(0.5 * logvar)                                   (vae.py:15-18)
__global__ __launch_bounds__(256) void reparam_fwd_k(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, float* __restrict__ z,
                                                     int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        z[i] = fmaf(eps[i], expf(0.5f * logvar[i]), mu[i]);
}

// dmu = dz [+ g_mu] [+ gs * mu];  dlogvar = dz * eps * 0.5 * exp(0.5 lv) [+ g_lv] [+ gs * 0.5 * (exp(lv) - 1)]
__global__ __launch_bounds__(256) void reparam_bwd_k(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, const float* __restrict__ dz,
                                                     const float* g_mu, const float* g_lv,
                                                     const float* __restrict__ gscale, int kl_grad, float* dmu,
                                                     float* dlv, int64_t n) {
    // g_mu/g_lv may alias dmu/dlv (autograd path adds the incoming gradient in place)
    const float gs = kl_grad ? (gscale != nullptr ? gscale[0] : 1.f) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float lv = logvar[i], d = dz[i];
        const float sd = expf(0.5f * lv);
        float a = d, b = d * eps[i] * 0.5f * sd;
        if (g_mu != nullptr) {
            a += g_mu[i];
            b += g_lv[i];
        }
        if (kl_grad) {
            a = fmaf(gs, mu[i], a);
            b = fmaf(gs, 0.5f * (sd * sd - 1.f), b);
        }
        dmu[i] = a;
        dlv[i] = b;
    }
}

// out[b][c] = in[b][c][0]                                              (Decoders.py:65)
__global__ __launch_bounds__(256) void slice0_fwd_k(const float* __restrict__ in, float* __restrict__ out,
                                                    int64_t rows, int L) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = in[i * L];
}
__global__ __launch_bounds__(256) void slice0_bwd_k(const float* __restrict__ dout, float* __restrict__ din,
                                                    int64_t total, int L) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        din[i] = (i % L == 0) ? dout[i / L] : 0.f;
}

// ---- ELBO ----------------------------------------------------------------------------------
__device__ __forceinline__ float bce_term(float p, float t) {
    // torch.nn.functional.binary_cross_entropy clamps each log at -100
    const float lp = fmaxf(logf(p), -100.f);
    const float l1 = fmaxf(logf(1.f - p), -100.f);
    return -(t * lp + (1.f - t) * l1);
}

// stage 1: per-block partial (recon, kld) sums -> partials[block][2]; grid = VAE_ELBO_BLOCKS
__global__ __launch_bounds__(256) void elbo_fwd_k(const float* __restrict__ xr, const float* __restrict__ tg,
                                                  const float* __restrict__ mu, const float* __restrict__ lv,
                                                  float* __restrict__ partials, int64_t B, int nps, int64_t t_sb,
                                                  int64_t t_se, int latent, int bce) {
    __shared__ float sm[4];
    float rec = 0.f, kl = 0.f;
    const int64_t nrec = B * nps, nkl = B * latent;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (t_se == 1 && t_sb == nps) {  // target laid out like x_rec: no index arithmetic (a 64-bit divide per
                                     // element costs more than the two logs)
        float rec2 = 0.f;
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + stride < nrec; i += 2 * stride) {
            const float p0 = xr[i], t0 = tg[i], p1 = xr[i + stride], t1 = tg[i + stride];
            if (bce) { rec += bce_term(p0, t0); rec2 += bce_term(p1, t1); }
            else { rec = fmaf(p0 - t0, p0 - t0, rec); rec2 = fmaf(p1 - t1, p1 - t1, rec2); }
        }
        if (i < nrec) {
            const float p = xr[i], t = tg[i];
            if (bce) rec += bce_term(p, t);
            else rec = fmaf(p - t, p - t, rec);
        }
        rec += rec2;
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += stride) {
            const int64_t b = i / nps;
            const int j = (int)(i - b * nps);
            const float p = xr[i], t = tg[b * t_sb + j * t_se];
            if (bce) rec += bce_term(p, t);
            else rec = fmaf(p - t, p - t, rec);
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nkl; i += stride) {
        const float m = mu[i], l = lv[i];
        kl += 1.f + l - m * m - expf(l);
    }
    const float r = block_sum_256(rec, sm);
    const float k = block_sum_256(kl, sm);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = r;
        partials[2 * blockIdx.x + 1] = -0.5f * k;
    }
}
// stage 2: fixed-order sum -> out[0]=loss, out[1]=recon, out[2]=kld; optionally out[3] += loss
__global__ __launch_bounds__(256) void elbo_finalize_k(const float* __restrict__ partials, float* __restrict__ out,
                                                       int nblocks, int accumulate) {
    __shared__ float sm[4];
    float r = 0.f, k = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
        r += partials[2 * i];
        k += partials[2 * i + 1];
    }
    r = block_sum_256(r, sm);
    k = block_sum_256(k, sm);
    if (threadIdx.x == 0) {
        out[0] = r + k;
        out[1] = r;
        out[2] = k;
        if (accumulate) out[3] += r + k;
    }
}

// d(loss)/d(x_rec) (and optionally the KL grads when dmu != nullptr), scaled by gscale[0]
// extra ranges the first kernel of a backward pass clears on its way (the BatchNorm accumulators: igemm.h BnFin)
#define ZERO_RANGES_MAX 8
struct ZeroRanges {
    float* p[ZERO_RANGES_MAX];
    int64_t n[ZERO_RANGES_MAX];
    int cnt;
};
__global__ __launch_bounds__(256) void elbo_bwd_k(const float* __restrict__ xr, const float* __restrict__ tg,
                                                  const float* __restrict__ mu, const float* __restrict__ lv,
                                                  const float* __restrict__ gscale, float* __restrict__ dxr,
                                                  float* __restrict__ dmu, float* __restrict__ dlv, int64_t B,
                                                  int nps, int64_t t_sb, int64_t t_se, int latent, int bce,
                                                  float* __restrict__ zero_buf, int64_t zero_n,
                                                  const float* __restrict__ fin_partials = nullptr,
                                                  float* __restrict__ fin_out = nullptr, int fin_blocks = 0,
                                                  int fin_accumulate = 0, ZeroRanges zr = ZeroRanges{}) {
    // fin_partials: the forward left the fixed-order sum of its partials to this launch (VAE_F_LOSS_IN_BWD: one graph
    // node less per train step); workgroup 0 does what elbo_finalize_k does
    if (fin_partials != nullptr && blockIdx.x == 0) {
        __shared__ float sm[4];
        float r = 0.f, k = 0.f;
        for (int i = threadIdx.x; i < fin_blocks; i += blockDim.x) {
            r += fin_partials[2 * i];
            k += fin_partials[2 * i + 1];
        }
        r = block_sum_256(r, sm);
        k = block_sum_256(k, sm);
        if (threadIdx.x == 0) {
            fin_out[0] = r + k;
            fin_out[1] = r;
            fin_out[2] = k;
            if (fin_accumulate) fin_out[3] += r + k;
        }
    }
    const float gs = gscale != nullptr ? gscale[0] : 1.f;
    const int64_t nrec = B * nps, nkl = B * latent;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // the first kernel of a backward pass also clears the flat gradient buffer (a memset node of its own costs a
    // launch: 4.6 us of a step that is bound by them)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += stride) zero_buf[i] = 0.f;
    // ... and the BatchNorm accumulators (igemm.h BnFin): this step's backward sums start from zero, and so do the next
    // step's forward statistics (a fused train step then needs no memset node at all)
    for (int q = 0; q < zr.cnt; ++q)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < zr.n[q]; i += stride) zr.p[q][i] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += stride) {
        int64_t ti = i;
        if (!(t_se == 1 && t_sb == nps)) {
            const int64_t b = i / nps;
            ti = b * t_sb + (i - b * nps) * t_se;
        }
        const float p = xr[i], t = tg[ti];
        float g;
        if (bce) g = (p - t) / fmaxf((1.f - p) * p, 1e-12f);  // torch's binary_cross_entropy_backward
        else g = 2.f * (p - t);
        dxr[i] = g * gs;
    }
    if (dmu != nullptr)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nkl; i += stride) {
            dmu[i] = gs * mu[i];
            dlv[i] = gs * 0.5f * (expf(lv[i]) - 1.f);
        }
}

// The ELBO of a train step in ONE launch (VAE_F_LOSS_IN_BWD, target laid out like x_rec): the loss terms and
// d(loss)/d(x_rec) come from the same pass over x_rec and the target (the two-launch form reads both twice: 38 MB
// instead of 26 MB per step at B = 4096, and a graph node more).  Grid and loop are those of elbo_fwd_k, so every
// workgroup's partial pair -- and with it the loss -- equals the two-launch form's bit for bit.  The fixed-order sum of
// the pairs is left to a LATER launch of the backward pass (wgrad_finalize_k's workgroup 0, else elbo_finalize_k): a
// last-workgroup-done ticket inside this kernel was measured at 38 us instead of 8 -- the device-scope release fence
// in front of the ticket writes the XCD's L2 back, and this kernel has just filled it with 12.8 MB of gradient.  Also
// what elbo_bwd_k does as the first kernel of a backward pass: it clears the flat gradient buffer and the BatchNorm
// accumulators.  The KL gradients are the REPARAM op's (VAE_F_KL_GRAD).
__global__ __launch_bounds__(256) void elbo_fused_k(const float* __restrict__ xr, const float* __restrict__ tg,
                                                    const float* __restrict__ mu, const float* __restrict__ lv,
                                                    const float* __restrict__ gscale, float* __restrict__ dxr,
                                                    float* __restrict__ partials, int64_t B, int nps, int latent,
                                                    int bce, float* __restrict__ zero_buf, int64_t zero_n,
                                                    ZeroRanges zr) {
    __shared__ float sm[4];
    const float gs = gscale != nullptr ? gscale[0] : 1.f;
    const int64_t nrec = B * nps, nkl = B * latent;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = i0; i < zero_n; i += stride) zero_buf[i] = 0.f;
    for (int q = 0; q < zr.cnt; ++q)
        for (int64_t i = i0; i < zr.n[q]; i += stride) zr.p[q][i] = 0.f;
    float rec = 0.f, rec2 = 0.f, kl = 0.f;
    auto grad = [&](float p, float t) {
        return (bce ? (p - t) / fmaxf((1.f - p) * p, 1e-12f) : 2.f * (p - t)) * gs;  // (elbo_bwd_k)
    };
    int64_t i = i0;
    for (; i + stride < nrec; i += 2 * stride) {
        const float p0 = xr[i], t0 = tg[i], p1 = xr[i + stride], t1 = tg[i + stride];
        if (bce) { rec += bce_term(p0, t0); rec2 += bce_term(p1, t1); }
        else { rec = fmaf(p0 - t0, p0 - t0, rec); rec2 = fmaf(p1 - t1, p1 - t1, rec2); }
        dxr[i] = grad(p0, t0);
        dxr[i + stride] = grad(p1, t1);
    }
    if (i < nrec) {
        const float p = xr[i], t = tg[i];
        if (bce) rec += bce_term(p, t);
        else rec = fmaf(p - t, p - t, rec);
        dxr[i] = grad(p, t);
    }
    rec += rec2;
    for (int64_t j = i0; j < nkl; j += stride) {
        const float m = mu[j], l = lv[j];
        kl += 1.f + l - m * m - expf(l);
    }
    const float r = block_sum_256(rec, sm);
    const float k = block_sum_256(kl, sm);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = r;
        partials[2 * blockIdx.x + 1] = -0.5f * k;
    }
}

// ---- Philox4x32-10 + Box-Muller ------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
}
// four standard normals per Philox counter base + q (Box-Muller on the four words): the body of randn_k, shared with
// adam_step_k's draw of the next step's eps
__device__ __forceinline__ void randn_quad(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t base, int64_t q) {
    const uint64_t ctr = base + (uint64_t)q;
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x5eed0001u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.f / 16777216.f), u1 = ((float)(c1 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.f / 16777216.f), u3 = ((float)(c3 >> 8) + 0.5f) * (1.f / 16777216.f);
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    float s0, cs0, s1, cs1;
    sincosf(6.2831853071795865f * u1, &s0, &cs0);
    sincosf(6.2831853071795865f * u3, &s1, &cs1);
    const float v[4] = {r0 * cs0, r0 * s0, r1 * cs1, r1 * s1};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * q + j < n) out[4 * q + j] = v[j];
}
// ---- Adam ----------------------------------------------------------------------------------
// tick block (device): {int32 step count t, int32 workgroups done, float 1-b1^t, float 1-b2^t}.  Every workgroup
// reads t and works with t+1; the last one to finish (a device-scope counter) stores t+1, so a captured graph
// replays with the right bias correction and the step counter needs no kernel of its own.
// hyper (optional, device): {lr, beta1, beta2, eps} read at run time instead of the by-value arguments, so a captured
// graph follows the optimiser's current learning rate (an LR scheduler between replays) without a re-capture.
// beta^t for the integer step count by repeated squaring in double: <= 31 rounds of two multiplications, where the
// general pow(double, double) is a few microseconds of code in front of every workgroup's loads (it was most of this
// kernel's 8 us on 100 k parameters); the product is within ~1e-15 of pow's, far below the float it is rounded to.
__device__ __forceinline__ double adam_ipow(double b, int t) {
    double r = 1.0;
    for (int e = t; e > 0; e >>= 1) {
        if (e & 1) r *= b;
        b *= b;
    }
    return r;
}
__global__ __launch_bounds__(256) void adam_step_k(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float beta1, float beta2, float eps, int32_t* tick,
                                                   const float* __restrict__ hyper, float* __restrict__ draw_out = nullptr,
                                                   int64_t draw_n = 0, uint64_t* __restrict__ rng = nullptr) {
    // draw_out: the NEXT step's reparameterisation noise is drawn here (randn_k's numbers from randn_k's state), so a
    // train step has no launch of its own for it; this is the step's last kernel, nothing reads the old noise any more
    __shared__ float sbc[2];
    uint64_t r_seed = 0, r_base = 0;
    if (draw_out != nullptr) { r_seed = rng[0]; r_base = rng[1]; }
    if (hyper != nullptr) {
        lr = hyper[0]; beta1 = hyper[1]; beta2 = hyper[2]; eps = hyper[3];
    }
    const int t = tick[0] + 1;
    // the first element's operands are requested before the bias corrections are worked out and shared (one latency
    // round for everything; most threads have exactly one element)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float gi = 0.f, m0 = 0.f, v0 = 0.f, p0 = 0.f;
    if (i < n) { gi = g[i]; m0 = m[i]; v0 = v[i]; p0 = p[i]; }
    if (threadIdx.x == 255) {
        sbc[0] = (float)(1.0 - adam_ipow((double)beta1, t));
        sbc[1] = (float)(1.0 - adam_ipow((double)beta2, t));
    }
    __syncthreads();
    const float step = lr / sbc[0];
    const float rs2 = rsqrtf(sbc[1]);  // torch: denom = sqrt(v)/sqrt(bc2) + eps
    while (i < n) {
        const float mi = fmaf(beta1, m0, (1.f - beta1) * gi);
        const float vi = fmaf(beta2, v0, (1.f - beta2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = p0 - step * mi / (sqrtf(vi) * rs2 + eps);
        i += stride;
        if (i < n) { gi = g[i]; m0 = m[i]; v0 = v[i]; p0 = p[i]; }
    }
    const int64_t nquad = (draw_n + 3) / 4;
    if (draw_out != nullptr) {
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride)
            randn_quad(draw_out, draw_n, r_seed, r_base, q);
        __syncthreads();  // (the ticket below is taken after every thread of this workgroup has drawn)
    }
    if (threadIdx.x == 0) {
        if (atomicAdd(&tick[1], 1) == (int)gridDim.x - 1) {  // every other workgroup has read tick[0] (and rng[1]) long ago
            float* bc = reinterpret_cast<float*>(tick) + 2;
            bc[0] = sbc[0];
            bc[1] = sbc[1];
            tick[1] = 0;
            tick[0] = t;
            if (draw_out != nullptr) rng[1] = r_base + (uint64_t)nquad;
        }
    }
}

// rng (device): {seed, offset, workgroups done}.  The offset is advanced by the last workgroup to finish (every
// other one has read it by then), so a draw is one launch.
__global__ __launch_bounds__(256) void randn_k(float* __restrict__ out, int64_t n, uint64_t* __restrict__ rng) {
    const uint64_t seed = rng[0], base = rng[1];
    const int64_t nquad = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x)
        randn_quad(out, n, seed, base, q);
    __syncthreads();  // (rng[1] was read by every thread of this workgroup)
    if (threadIdx.x == 0) {
        unsigned long long* done = reinterpret_cast<unsigned long long*>(rng + 2);
        if (atomicAdd(done, 1ull) == (unsigned long long)gridDim.x - 1) {
            rng[2] = 0;
            rng[1] = base + (uint64_t)nquad;
        }
    }
}
// ---- sliding windows of a resident series ---------------------------------------------------
// out[b][c][l] = series[c][starts[b] + l]: slidingWindow.__getitem__ (train.py:25-28) for a whole batch on the
// device; the DataLoader's collate + host-to-device copy of B*C*L floats becomes a gather from the (C, T) series
// that stays in HBM (windows overlap by L-1 samples, so the series is L times smaller than an epoch of windows).
// A start the reference would answer with None (start + L > T, or negative) yields NaNs.
__global__ __launch_bounds__(256) void window_gather_k(const float* __restrict__ series,
                                                       const int64_t* __restrict__ starts, float* __restrict__ out,
                                                       int C, int64_t T, int L, int B) {
    const int per = C * L;
    const int64_t total = (int64_t)B * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per;
        const int r = (int)(i - b * per);
        const int c = r / L, l = r - c * L;
        const int64_t s = starts[b];
        out[i] = (s >= 0 && s + L <= T) ? series[(int64_t)c * T + s + l] : __builtin_nanf("");
    }
}
