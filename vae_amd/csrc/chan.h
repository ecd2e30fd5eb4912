// chan.h - per-channel reductions and BatchNorm kernels over [B][C][HW] fp32 tensors.
//
// Two-stage deterministic reductions: stage 1 (chan_reduce_k) has one block per
// (channel, batch-chunk) writing its partial pair to partials[chunk][c][2]; stage 2 (the
// *_finalize kernels) sums the chunks in a fixed order.  No float atomics, so BatchNorm
// statistics and bias gradients are bit-stable run to run.
#pragma once
#include "common.h"

enum { CR_STATS = 0, CR_ACTBWD = 1, CR_BNBWD = 2 };

// CR_STATS : s1 = sum(x - pivot[c]), s2 = sum((x - pivot[c])^2)      (pivot = x[0,c,0], for
//            numerically robust variance when |mean| >> std)
// CR_ACTBWD: g = dy * act'(y); dy <- g (in place); s1 = sum(g)       (bias gradient)
// CR_BNBWD : g = dout * (relu ? out > 0 : 1); dout <- g; s1 = sum(g); s2 = sum(g * xhat)
// A chunk holds < 2^31 elements (host-checked), so the element loop runs on 32-bit indices: the 64-bit
// division the flat index would need costs more than the loads.
template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_k(const float* __restrict__ x, float* __restrict__ dy,
                                                     const float* __restrict__ y, const float* __restrict__ save,
                                                     float* __restrict__ partials, int B, int C, int HW, int bchunk,
                                                     int act, float slope, int relu) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    const int b0 = blockIdx.y * bchunk;
    const int b1 = min(B, b0 + bchunk);
    float s1 = 0.f, s2 = 0.f;
    float pivot = 0.f, mean = 0.f, rstd = 0.f;
    if (MODE == CR_STATS) pivot = x[(int64_t)c * HW];
    if (MODE == CR_BNBWD) {
        mean = save[c];
        rstd = save[C + c];
    }
    const int n = (b1 - b0) * HW;
    const int64_t base = ((int64_t)b0 * C + c) * HW;  // element (b0, c, 0)
    const int skip = (C - 1) * HW;                    // added per sample crossed
    auto one = [&](int q) {
        const int bb = (int)((unsigned)q / (unsigned)HW);
        const int64_t idx = base + q + (int64_t)bb * skip;
        if (MODE == CR_STATS) {
            const float v = x[idx] - pivot;
            s1 += v;
            s2 = fmaf(v, v, s2);
        } else if (MODE == CR_ACTBWD) {
            float g = dy[idx];
            if (act != VAE_ACT_NONE) {
                g *= act_grad_from_out(y[idx], act, slope);
                dy[idx] = g;
            }
            s1 += g;
        } else {
            float g = dy[idx];
            if (relu) {
                g = y[idx] > 0.f ? g : 0.f;
                dy[idx] = g;
            }
            const float xh = (x[idx] - mean) * rstd;
            s1 += g;
            s2 = fmaf(g, xh, s2);
        }
    };
    int q = threadIdx.x;
    for (; q + 768 < n; q += 1024) {  // four independent element streams in flight
        one(q);
        one(q + 256);
        one(q + 512);
        one(q + 768);
    }
    for (; q < n; q += 256) one(q);
    const float t1 = block_sum_256(s1, sm);
    const float t2 = block_sum_256(s2, sm);
    if (threadIdx.x == 0) {
        float* p = partials + ((int64_t)blockIdx.y * C + c) * 2;
        p[0] = t1;
        p[1] = t2;
    }
}

// BatchNorm train-mode statistics -> save[0:C]=mean, save[C:2C]=rstd; running stats updated
// like torch: running = (1-m)*running + m*stat, with the UNBIASED variance for running_var.
__global__ void bn_stats_finalize_k(const float* __restrict__ x, const float* __restrict__ partials,
                                    float* __restrict__ save, float* __restrict__ rmean, float* __restrict__ rvar,
                                    int C, int HW, int nchunks, int64_t n, float eps, float momentum,
                                    float* __restrict__ row_out) {
    // one 64-lane workgroup per channel: the lanes take the chunks, a fixed shuffle tree adds them
    const int c = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < nchunks; k += 64) {
        s1 += partials[((int64_t)k * C + c) * 2];
        s2 += partials[((int64_t)k * C + c) * 2 + 1];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (threadIdx.x != 0) return;
    const float pivot = x[(int64_t)c * HW];
    const float inv_n = 1.f / (float)n;
    const float dm = s1 * inv_n;
    const float mean = pivot + dm;
    if (row_out != nullptr) {  // data-parallel: this rank's {mean, -, M2, n} row (merged by bnv_finalize_fwd_k)
        row_out[c * 4] = mean;
        row_out[c * 4 + 1] = 0.f;
        row_out[c * 4 + 2] = fmaxf(s2 - s1 * dm, 0.f);
        row_out[c * 4 + 3] = (float)n;
        return;
    }
    float var = s2 * inv_n - dm * dm;
    var = var > 0.f ? var : 0.f;
    save[c] = mean;
    save[C + c] = rsqrtf(var + eps);
    if (rmean != nullptr) {
        const float unbiased = n > 1 ? var * ((float)n / (float)(n - 1)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
}

// Eval mode: normalise with the running statistics.
__global__ void bn_eval_prepare_k(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                  float* __restrict__ save, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save[c] = rmean[c];
    save[C + c] = rsqrtf(rvar[c] + eps);
}

__global__ __launch_bounds__(256) void bn_apply_k(const float* __restrict__ x, float* __restrict__ y,
                                                  const float* __restrict__ save, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int64_t total, int C, int HW,
                                                  int relu) {
    const bool small = total < ((int64_t)1 << 31);  // 32-bit index arithmetic when the tensor allows it
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = small ? (int)(((unsigned)i / (unsigned)HW) % (unsigned)C) : (int)((i / HW) % C);
        const float sc = save[C + c] * gamma[c];
        float v = fmaf(x[i] - save[c], sc, beta[c]);
        if (relu) v = v > 0.f ? v : 0.f;
        y[i] = v;
    }
}

// sum over chunks -> dgamma = sum(g*xhat), dbeta = sum(g); save[2C:3C] = sum(g)/n, save[3C:4C] = sum(g*xhat)/n
__global__ void bn_bwd_finalize_k(const float* __restrict__ partials, float* __restrict__ save,
                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int nchunks,
                                  int64_t n, float* __restrict__ row_out) {
    const int c = blockIdx.x;  // one 64-lane workgroup per channel, as in bn_stats_finalize_k
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < nchunks; k += 64) {
        s1 += partials[((int64_t)k * C + c) * 2];
        s2 += partials[((int64_t)k * C + c) * 2 + 1];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (threadIdx.x != 0) return;
    if (row_out != nullptr) {  // data-parallel: this rank's {sum g, sum g*xhat, n} row
        row_out[c * 4] = s1;
        row_out[c * 4 + 1] = s2;
        row_out[c * 4 + 2] = (float)n;
        row_out[c * 4 + 3] = 0.f;
        return;
    }
    dbeta[c] = s1;
    dgamma[c] = s2;
    save[2 * C + c] = s1 / (float)n;
    save[3 * C + c] = s2 / (float)n;
}

// Data-parallel second pass: rows[rank][c] = {sum g, sum g*xhat, n} of every rank.  The means are global,
// dgamma/dbeta stay this rank's local sums (the gradient all-reduce adds the ranks).
__global__ void bn_bwd_rows_k(const float* __restrict__ rows, int world, const float* __restrict__ local_row,
                              float* __restrict__ save, float* __restrict__ dgamma, float* __restrict__ dbeta,
                              int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f, n = 0.f;
    for (int r = 0; r < world; ++r) {
        s1 += rows[((int64_t)r * C + c) * 4];
        s2 += rows[((int64_t)r * C + c) * 4 + 1];
        n += rows[((int64_t)r * C + c) * 4 + 2];
    }
    dbeta[c] = local_row[c * 4];
    dgamma[c] = local_row[c * 4 + 1];
    save[2 * C + c] = s1 / n;
    save[3 * C + c] = s2 / n;
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g*xhat));  g already ReLU-masked in dout.
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const float* __restrict__ x, const float* __restrict__ g,
                                                      float* __restrict__ dx, const float* __restrict__ save,
                                                      const float* __restrict__ gamma, int64_t total, int C, int HW) {
    const bool small = total < ((int64_t)1 << 31);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = small ? (int)(((unsigned)i / (unsigned)HW) % (unsigned)C) : (int)((i / HW) % C);
        const float rstd = save[C + c];
        const float xh = (x[i] - save[c]) * rstd;
        dx[i] = gamma[c] * rstd * (g[i] - save[2 * C + c] - xh * save[3 * C + c]);
    }
}

// bias gradient finalize: db[c] = sum over chunks of partial s1
__global__ void bias_grad_finalize_k(const float* __restrict__ partials, float* __restrict__ db, int C,
                                     int nchunks) {
    const int c = blockIdx.x;  // one 64-lane workgroup per channel
    float s = 0.f;
    for (int k = threadIdx.x; k < nchunks; k += 64) s += partials[((int64_t)k * C + c) * 2];
    s = wave_sum(s);
    if (threadIdx.x == 0) db[c] = s;
}
