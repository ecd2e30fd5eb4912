/* vae_amd.h - C ABI of libvae_amd.so: the MI355X (gfx950) VAE training hot path.
 *
 * The reference (AnisKadri/VAE) has no FFI: its boundary for this path is the Python
 * nn.Module duck type.  This library is what sits underneath the drop-in
 * Python modules in vae_amd/: plain pointers, sizes and a hipStream_t, no torch types.
 * Every entry point returns 0 on success or a negative VAE_E* code; vae_last_error()
 * returns a human-readable message for the calling thread's last failure.  All pointers are
 * DEVICE pointers to fp32 unless stated; all launches go to the given stream and nothing
 * synchronises or allocates (safe to capture in a hipGraph), except vae_graph_* which own
 * graph objects.
 *
 * The hot path is expressed as a PROGRAM: an array of vae_op records executed in order by
 * vae_run_forward and in reverse by vae_run_backward.  One op = one fused layer primitive
 * of the reference model:
 *
 *   kind          replaces (reference file:line)
 *   VAE_OP_CONV   nn.Conv1d / nn.Conv2d + LeakyReLU/ReLU that follows it
 *                 (Encoders.py:42-43,46-47; multiscale_vae.py:54-63,102-111;
 *                  vae-checkpoint.ipynb:37-43)
 *   VAE_OP_CONVT  nn.ConvTranspose1d / 2d + activation / sigmoid
 *                 (Decoders.py:49-53; multiscale_vae.py:172-181; vae-checkpoint.ipynb:91-104)
 *   VAE_OP_BN     nn.BatchNorm1d / 2d, train (batch stats, running-stat update) or eval,
 *                 optional fused ReLU (Encoders.py:44,48; Decoders.py:54; notebook :40-41,92-96)
 *   VAE_OP_LINEAR nn.Linear (+ReLU) (Encoders.py:51-58,84-85; Decoders.py:41,59; notebook :48-57,84-86)
 *   VAE_OP_REPARAM reparametrization_trick (vae.py:15-18; multiscale_vae.py:206-209)
 *   VAE_OP_SLICE0 x[:, :, 0] (Decoders.py:65; multiscale_vae.py:192)
 *   VAE_OP_ELBO   criterion: MSE-sum or BCE-sum reconstruction + analytic KL
 *                 (train.py:39-48; vae-checkpoint.ipynb:141-151)
 *
 * Buffer references ("refs") are int64: (space << 56) | offset_in_floats.  `space` indexes
 * the `bases` pointer table handed to vae_run_*; by convention of the Python host:
 *   0 WORK (activations + activation grads)   1 PARAM (flat parameters)
 *   2 GRAD (flat parameter grads, same offsets as PARAM)   3 STATE (BN running stats, rng)
 *   4.. EXT (caller tensors: x, eps, outputs, incoming grads)
 * A ref of -1 means "absent".  Extents are validated on the host against `base_sizes`
 * before any launch.
 */
#ifndef VAE_AMD_H
#define VAE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vae_stream_t; /* == hipStream_t */

enum {
    VAE_OK = 0,
    VAE_EINVAL = -1,   /* bad argument / shape mismatch / out-of-bounds ref */
    VAE_EHIP = -2,     /* a HIP runtime call failed */
    VAE_ENODEV = -3,   /* no gfx950 device */
};

enum { VAE_OP_CONV = 1, VAE_OP_CONVT = 2, VAE_OP_BN = 3, VAE_OP_LINEAR = 4,
       VAE_OP_REPARAM = 5, VAE_OP_SLICE0 = 6, VAE_OP_ELBO = 7 };
enum { VAE_ACT_NONE = 0, VAE_ACT_LEAKY = 1 /* slope in f[0]; 0 => ReLU */, VAE_ACT_SIGMOID = 2 };
enum {
    VAE_F_ACC_DIN = 1,     /* LINEAR/CONV: input grad is accumulated (+=), input has >1 consumer */
    VAE_F_BN_RELU = 2,     /* BN: ReLU fused after normalisation (notebook Conv->BN->ReLU) */
    VAE_F_KL_GRAD = 4,     /* REPARAM bwd: add d(KL)/dmu, d(KL)/dlogvar scaled by the ELBO op's g */
    VAE_F_EXT_GRAD = 8,    /* REPARAM bwd: add incoming grads r[7], r[8] (autograd path) */
    VAE_F_BCE = 16,        /* ELBO: BCE-sum reconstruction (else MSE-sum) */
    VAE_F_IN_RELU = 32,    /* CONV/CONVT fused: ReLU follows the producer's (virtual) BatchNorm */
    VAE_F_FUSED = 64,      /* CONV/CONVT: LDS-staged MFMA implicit-GEMM kernels; BN: "virtual" (its output is
                              never materialised: the producer's epilogue takes the statistics, the consumer
                              normalises on load, and the backward pass is folded the same way) */
    VAE_F_LOSS_IN_BWD = 128, /* ELBO: the caller promises that the op's backward runs behind every forward (a train
                              step).  Target laid out like x_rec (d[2] == 1, d[1] == d[0]): forward launches nothing,
                              backward computes the loss terms and the gradient in one pass (elbo_fused_k) and the
                              fixed-order sum into out[0..3] rides on the weight-gradient reduction that ends the
                              backward call.  Otherwise forward leaves the per-workgroup partials and the op's backward
                              launch adds them.  Either way out[0..3] is valid after the backward call, not before.
                              No counterpart in the reference (train.py:39-48 computes the loss eagerly). */
    VAE_F_BN_ACC = 256,    /* BN (VAE_F_FUSED): the save block r[6] is followed by VAE_BN_ACC_FLOATS(C) floats of
                              fixed-point accumulators (the producers' per-channel sums, added with 64-bit integer
                              atomics), and the save blocks of all such BN ops of a program lie back to back (each padded
                              to a multiple of 64 floats) so that one memset at the start of a training forward clears
                              them.  The BatchNorm is then finalised in the prologue of its first consumer kernel
                              instead of a launch of its own.  CONV/CONVT (VAE_F_FUSED): the blocks r[7] / r[8] name
                              are such blocks.  Single GPU only (under vae_set_collective the rows are gathered). */
};
#define VAE_BN_ACC_FLOATS(C) (1536 * (C))  /* 2 directions x 3 quantities x 16 shards x C channels x a 64-byte line each, in floats */

/* One program op.  Field use per kind ("-" unused):
 *  CONV/CONVT r: 0 in, 1 out, 2 din(-1: none), 3 dout, 4 w, 5 b (w,b: PARAM refs), 6 partials (WORK,
 *                2*Cout*VAE_MAX_CHUNKS floats of scratch for the bias-gradient reduction)
 *             d: 0 Cin 1 H 2 W 3 Cout 4 OH 5 OW 6 KH 7 KW 8 sh 9 sw 10 ph 11 pw
 *             act, f[0]=slope.  Weight layout: CONV [Cout][Cin][KH][KW], CONVT [Cin][Cout][KH][KW].
 *             VAE_F_FUSED adds: 7 in_bn (save block of the virtual BN that produced `in`, -1 none; then
 *             `in`/`din` name that BN's INPUT and its gradient), 8 out_bn (save block of the virtual BN
 *             that follows, -1 none), 9 running_mean of that BN (statistics pivot); r6 then holds the
 *             per-chunk statistics slab (vae_conv_scratch_floats), 10 the op's own weight-gradient
 *             partial region (vae_conv_wgrad_partial_floats; backward only).  Without VAE_F_FUSED r[10] is
 *             optional: a wide layer (>= 16 channels on the small side of its weight gradient) uses it for
 *             split-reduction partials when present and falls back to float atomics when -1.
 *  BN         r: 0 in, 1 out, 2 din, 3 dout, 4 gamma, 5 beta, 6 save (WORK, 8*C floats: mean, rstd,
 *                scale|sum_dy/n, shift|sum_dy_xhat/n, a, b, c, spare), 7 running_mean, 8 running_var
 *                (STATE), 9 partials / slab (WORK, shared scratch).  VAE_F_FUSED: 1 and 3 are -1.
 *             d: 0 C, 1 HW.   f[0]=eps f[1]=momentum
 *  LINEAR     r: 0 in, 1 out, 2 din(-1), 3 dout, 4 w [N][K], 5 b, 6 partials (WORK, 2*N*VAE_MAX_CHUNKS),
 *                10 the op's own weight-gradient partial region (WORK, vae_linear_wgrad_partial_floats; backward only)
 *             d: 0 K, 1 N, 2 ld_in, 3 ld_out, 4 ld_din, 5 ld_dout.   act NONE or LEAKY(0)
 *  REPARAM    r: 0 mu, 1 logvar, 2 eps, 3 z, 4 dz, 5 dmu(out), 6 dlogvar(out), 7 g_mu, 8 g_logvar
 *                (EXT_GRAD), 9 gscale (device float, KL_GRAD; -1 => 1.0)
 *             d: 0 latent (row length; mu/logvar/eps/z contiguous [B][latent])
 *  SLICE0     r: 0 in [B][C][L], 1 out [B][C], 2 din, 3 dout.   d: 0 C, 1 L
 *  ELBO       r: 0 x_rec, 1 target, 2 mu, 3 logvar, 4 out (3 floats: loss, recon, kld; plus
 *                out[3] += loss when d[5]), 5 dx_rec, 6 partials (WORK, 2*VAE_ELBO_BLOCKS floats),
 *                7 gscale (device float, -1 => 1.0)
 *             d: 0 n_per_sample, 1 target batch stride, 2 target element stride, 3 latent,
 *                4 - , 5 accumulate running loss into out[3]
 */
typedef struct vae_op {
    int32_t kind;
    int32_t act;
    int32_t flags;
    int32_t reserved;
    int64_t r[12];
    int32_t d[16];
    float f[4];
} vae_op;

#define VAE_MAX_CHUNKS 64
#define VAE_ELBO_BLOCKS 1024

const char* vae_last_error(void);
/* Queries device 0..: returns 0 if the current device is gfx950, VAE_ENODEV otherwise. */
int vae_check_device(void);

/* Executes ops[0..n) forward / backward for a batch of B samples.
 * bases[i] / base_sizes[i] (floats) describe the address spaces; training != 0 selects
 * BatchNorm batch statistics + running-stat update.  training == 1 also clears the BatchNorm accumulator blocks
 * (VAE_F_BN_ACC) first; training == 2 skips that: the caller vouches that they are zero (a zeroed workspace, or the
 * previous pass over this workspace was a vae_run_backward* that began with the program's ELBO op, which clears them
 * on the way: the fused train step). */
int vae_run_forward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                    int n_bases, int B, int training, vae_stream_t stream);
int vae_run_backward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                     int n_bases, int B, vae_stream_t stream);
/* vae_run_backward followed by the Adam update (train.py:51-53: loss.backward(); optimizer.step()) of the PARAM space
 * bases[1] from the GRAD space bases[2]: m, v = Adam's moments, hyper = device {lr, beta1, beta2, eps}, tick as in
 * vae_adam_step_hyper.  One host call for everything behind the forward pass of a train step.  next_eps (optional,
 * with rng as in vae_randn): the Adam launch, the last of the step, also draws the next_eps_n standard normals of the
 * NEXT step's reparameterisation (vae.py:15-18 draws them with randn_like at every forward), the same numbers from the
 * same state that a vae_randn call at this point would give: a step then has no launch of its own for the noise. */
int vae_run_backward_step(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes, int n_bases,
                          int B, vae_stream_t stream, float* adam_m, float* adam_v, const float* hyper, int32_t* tick,
                          float* next_eps, int64_t next_eps_n, uint64_t* rng);
/* Backward over ops[op_begin, op_end) only (in reverse), with every weight gradient of the range complete when
 * the call's work is: a data-parallel host runs the decoder's range first, starts the all-reduce of that bucket
 * of the GRAD buffer on a side stream, and runs the encoder's range meanwhile (no counterpart in the reference,
 * which is single-device: SURVEY.md 8e).  zero_grads != 0 clears the GRAD space first (the first range of a
 * step); vae_run_backward == range [0, n_ops) with zero_grads = 1. */
int vae_run_backward_range(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                           int n_bases, int B, vae_stream_t stream, int op_begin, int op_end, int zero_grads);
/* Profiling twin of vae_run_forward: a HIP event is recorded on `stream` before every op and after the
 * last one; op_ms[i] (n_ops floats, host memory) receives the device time of op i in milliseconds.
 * Synchronises the stream, so it cannot be captured.  bench.py measures the dominant kernel with it. */
int vae_time_forward(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes,
                     int n_bases, int B, int training, vae_stream_t stream, float* op_ms);

/* Profiling twin of forward + backward for a fused 1-D program (the one vae_seq_scratch_floats accepted): a HIP event
 * is recorded on `stream` before every launch of the phase kernels.  ms[k] = device time of launch k in milliseconds,
 * bytes[k] = its algorithmic bytes (each input read once, each output written once, weights included), tags = 8
 * chars per launch ("fe0".., "fh", "fd1".., "elbofin", "bd2".., "bh", "be2".., "reduce"); cap = room in the three
 * arrays, *n_launches = launches written.  Needs the GRAD space bound.  Synchronises the stream; not capturable.
 * VAE_EINVAL when the program does not run on that path.  bench.py's roofline for the 1-D models comes from it. */
int vae_time_seq(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes, int n_bases, int B,
                 vae_stream_t stream, float* ms, double* bytes, char* tags, int cap, int* n_launches);

/* Profiling twin of a whole train step (train.py:57-66: forward, criterion, backward, optimizer.step): the launches of
 * vae_run_forward(training) followed by vae_run_backward_step (hyper != NULL) or vae_run_backward (hyper == NULL), with
 * a HIP event recorded on `stream` behind EVERY kernel launch.  For launch k: ms[k] = device time between the events
 * around it, names + 64*k = its kernel name as rocprofv3 prints it (template arguments included; fused 1-D phases also
 * carry their phase tag), bytes[k] = algorithmic bytes (every input read once, every output written once, weights
 * included), flops[k] = 2 x multiply-adds for the matrix-bound wide kernels (0 elsewhere).  cap = room in the arrays
 * (names: 64 * cap chars), *n_launches = launches written.  The step is REAL (parameters, Adam state and BatchNorm
 * statistics advance).  Synchronises the stream; not capturable.  bench.py derives its roofline line from it: the
 * kernel family with the largest total time over the step. */
int vae_time_step(const vae_op* ops, int n_ops, float* const* bases, const int64_t* base_sizes, int n_bases, int B,
                  vae_stream_t stream, float* adam_m, float* adam_v, const float* hyper, int32_t* tick, float* ms,
                  double* bytes, double* flops, char* names, int cap, int* n_launches);

/* Compute dtype of the matrix kernels (convolutions on the fused and wide paths, Linear layers), process-global.
 * VAE_DTYPE_F32 (default): fp32 MFMA, results within the fp32 tolerance of the reference.  VAE_DTYPE_BF16: the MFMA
 * operands are rounded to bf16 (round to nearest even) as they leave LDS, products accumulate in fp32; activations,
 * parameters, gradients, BatchNorm statistics, sigmoid/ELBO and Adam stay fp32 in memory and arithmetic (the
 * "fp32 then bf16" variant of BASELINE configs[1]; the reference itself is fp32 only, train.py:57-67).  Thin layers
 * that run on the vector ALU (<= 8 output channels with a short reduction) compute in fp32 in both modes, and a
 * fused 1-D program (vae_seq_scratch_floats) runs op by op under VAE_DTYPE_BF16.  Do not change it between the
 * forward and the backward of one step, nor while a captured graph of the other dtype is replayed (a graph keeps the
 * kernels it was captured with). */
#define VAE_DTYPE_F32 0
#define VAE_DTYPE_BF16 1
int vae_set_compute_dtype(int dtype);
int vae_get_compute_dtype(void);

/* Data parallelism (one process per GPU).  The reference is single-device: with equal shards,
 * gradient all-reduce SUM (done by the host on the flat GRAD buffer) plus synchronised BatchNorm statistics
 * reproduce its result at the global batch.  For the latter the executor calls `fn` at every training-mode
 * BatchNorm, forward and backward: buf holds `world` rows of floats_per_rank floats, row `rank` is filled;
 * fn must all-gather in place, ordered after the work already enqueued on `stream` and before what is
 * enqueued next (torch.distributed on the current stream does that).  Returns 0 on success.
 * world <= 1 or fn == NULL switches back to single-GPU statistics.  Process-global; not capturable. */
#define VAE_MAX_WORLD 64
typedef int (*vae_allgather_fn)(void* user, float* buf, int64_t floats_per_rank, vae_stream_t stream);
int vae_set_collective(vae_allgather_fn fn, void* user, int rank, int world);

/* The same two exchanges done by the library itself over RCCL, enqueued on the caller's stream: no callback into the
 * host language between the kernels of a step, so a data-parallel step is one stream-ordered sequence (capturable).
 * No counterpart in the reference (single device).  librccl is opened at run time from `rccl_path`
 * (NULL/"": the loader's search path), so a single-GPU user never loads it.
 *   vae_dp_unique_id       : rank 0 makes the VAE_DP_ID_BYTES id that every rank passes to vae_dp_init (two
 *                            ncclUniqueIds: one communicator for the SyncBN all-gathers on the compute stream, one for
 *                            the gradient all-reduces that run beside them on the caller's communication stream);
 *                            the caller ships it to the other ranks (any side channel: torch.distributed, MPI, a file).
 *   vae_dp_init            : this rank joins both communicators on the CURRENT device (ncclCommInitRank: collective,
 *                            every rank calls it).  One pair per process.
 *   vae_dp_enable(1)       : installs the library's own all-gather as the SyncBN hook (as vae_set_collective would)
 *                            with the communicator's rank / world; vae_dp_enable(0) removes it.
 *   vae_dp_all_gather_rows : the hook's exchange on its own: buf = `world` rows of floats_per_rank floats, row `rank`
 *                            filled; in-place ncclAllGather on `stream`.
 *   vae_dp_all_reduce_sum  : in-place ncclAllReduce(SUM) of n floats on `stream` (the flat gradient buffer or a
 *                            bucket of it; the ELBO is a sum over samples, so gradients add).
 *   vae_dp_world           : world size of the communicator, 0 when there is none.
 *   vae_dp_shutdown        : destroys the communicator (and removes the hook if it is the installed one).
 *   vae_dp_set_single_comm : (before vae_dp_init) 1 = join ONE communicator only: vae_dp_all_reduce_sum then uses the
 *                            SyncBN communicator, and the caller must enqueue every collective on ONE stream, so no
 *                            two RCCL kernels of this library are ever in flight together.  The fallback for systems
 *                            where two communicators used concurrently stall; default 0 (two communicators, gradient
 *                            buckets overlapped with the encoder's backward).  vae_dp_single_comm returns the setting. */
#define VAE_DP_ID_BYTES 256
int vae_dp_unique_id(const char* rccl_path, void* id256);
int vae_dp_init(const char* rccl_path, const void* id256, int rank, int world);
int vae_dp_world(void);
int vae_dp_enable(int on);
int vae_dp_all_gather_rows(float* buf, int64_t floats_per_rank, vae_stream_t stream);
int vae_dp_all_reduce_sum(float* buf, int64_t n, vae_stream_t stream);
int vae_dp_shutdown(void);
int vae_dp_set_single_comm(int on);
int vae_dp_single_comm(void);

/* 1 if the CONV/CONVT op can run on the fused LDS-staged kernels at batch B (host-only query used by
 * the Python program builder; no GPU needed). */
int vae_conv_fused_ok(const vae_op* op, int B);
/* Which instance of the fused forward kernel the op gets at batch B (host-only; bench.py groups its launches
 * by kernel name with it): 0 = not fused, 300 = the streaming kernels of the single-channel outer layers (thin_conv_fwd_k /
 * thin_convT_fwd_k), 400 = the streaming kernel of mid-size plain convolutions (mid_conv_k), else 100*NT + ND as in
 * igemm_gather_k<NT, 2, false, ND>. */
int vae_conv_forward_variant(const vae_op* op, int B);
/* Floats of shared scratch (the op's r[6] slot) a FUSED CONV/CONVT op needs at batch B for its BatchNorm statistic
 * slabs (+ the gathered data-parallel rows).  0 if not fusable. */
int64_t vae_conv_scratch_floats(const vae_op* op, int B);
/* Floats of the op's OWN weight-gradient partial region (r[10], WORK): one row per workgroup of the weight-gradient
 * kernel (fused path) or one copy of dw per slice of the reduction (wide layers off the fused path; 0 for layers
 * that need none).  The regions of different layers must not overlap: their reduction is deferred to one launch at the
 * end of vae_run_backward. */
int64_t vae_conv_wgrad_partial_floats(const vae_op* op, int B);
/* Floats of a LINEAR op's OWN weight-gradient partial region (r[10], WORK): one row of N*K + N floats per batch
 * slice of the weight-gradient kernel.  The rows are added in a fixed order by the same launch that reduces the conv
 * layers' partials, so Linear gradients are bit-identical from run to run (the reference's autograd on CPU is
 * deterministic too: train.py:57-67).  Backward of a LINEAR op without the region fails with VAE_EINVAL. */
int64_t vae_linear_wgrad_partial_floats(const vae_op* op, int B);

/* Whole-model path of the 1-D families (TCVAE: Encoders.py:76-107 + Decoders.py:57-78 + vae.py:20-23;
 * multiscale_vae.py:48-220).  When the op array is a complete 1-D VAE program -- nenc x {N x [CONV 1xk stride 2, BN],
 * LINEAR mu, LINEAR logvar}, REPARAM, LINEAR, N x [CONVT 1x2 stride 2, (BN)], SLICE0, [ELBO] -- vae_run_forward /
 * vae_run_backward execute it as one launch per BatchNorm-delimited phase (2N forward, 2N backward, one gradient
 * reduction) instead of several launches per layer; results are the same ops' results.  Returns the floats of WORK
 * scratch that path needs at batch B for a flat parameter buffer of nflat floats, or 0 when the program does not
 * match (host-only query).  The host stores the scratch reference in ops[0].r[11]; with r[11] = -1, under
 * vae_set_collective (world > 1) or with VAE_AMD_SEQ1D=0 in the environment the program runs op by op.  A fused
 * program's backward runs whole: vae_run_backward_range accepts only [0, n_ops) for it. */
int64_t vae_seq_scratch_floats(const vae_op* ops, int n_ops, int B, int64_t nflat);

/* Standalone criterion (autograd path).  Replaces train.py:39-48 / notebook :141-151.
 * out: 3 floats (loss, recon, kld).  partials: >= 2*VAE_ELBO_BLOCKS floats of scratch.
 * Target element (b, j) is at target[b*t_sb + j*t_se]. */
int vae_elbo_forward(const float* x_rec, const float* target, const float* mu, const float* logvar,
                     float* out, float* partials, int B, int n_per_sample, int64_t t_sb, int64_t t_se,
                     int latent, int bce, vae_stream_t stream);
int vae_elbo_backward(const float* x_rec, const float* target, const float* mu, const float* logvar,
                      const float* gscale, float* dx_rec, float* dmu, float* dlogvar, int B,
                      int n_per_sample, int64_t t_sb, int64_t t_se, int latent, int bce,
                      vae_stream_t stream);
/* vae_elbo_forward for callers that cannot allocate (code under hipGraph capture, tight serving loops, harnesses
 * that forbid hipMalloc): NO counterpart in the reference.  Same kernels, same result bit for bit, but the
 * partial-sum scratch is a static 8 KiB device array inside the library (2*VAE_ELBO_BLOCKS floats per GPU, held
 * until the process ends) instead of the `partials` argument; the call allocates and frees nothing.  Inputs of more
 * than VAE_ELBO_NOALLOC_MAX_ELEMS reconstruction (B*n_per_sample) or latent (B*latent) elements are rejected with
 * VAE_EINVAL.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams) race;
 * calls on one stream are ordered and need no clearing in between (every word read is rewritten by the same call). */
#define VAE_ELBO_NOALLOC_MAX_ELEMS ((int64_t)1 << 31)
int vae_elbo_forward_noalloc(const float* x_rec, const float* target, const float* mu, const float* logvar,
                             float* out, int B, int n_per_sample, int64_t t_sb, int64_t t_se, int latent, int bce,
                             vae_stream_t stream);

/* Fused flat Adam (torch.optim.Adam defaults semantics; draft_pipe.py:132, notebook :249).
 * tick: device int32[4] block {step count t, scratch (workgroups done: keep 0), float 1-b1^t, float 1-b2^t}, zeroed
 * by the caller before the first step; the kernel works with t+1 and stores it (graph-replay safe). */
int vae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int32_t* tick, vae_stream_t stream);
/* Same step with the hyper-parameters read from DEVICE memory at run time: hyper = float[4] {lr, beta1, beta2,
 * eps}.  A captured graph of the train step then follows `optimizer.param_groups[0]['lr']` (LR schedulers,
 * manual decay between epochs: the reference's drivers set lr once, draft_pipe.py:132, but torch.optim.Adam honours
 * later changes) after a 16-byte refresh of `hyper`, without a re-capture. */
int vae_adam_step_hyper(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                        int32_t* tick, vae_stream_t stream);

/* eps ~ N(0,1): Philox4x32-10 + Box-Muller; rng = device uint64[3] {seed, offset, 0 (scratch: keep 0)}; offset is
 * advanced on the device (replaces torch.randn_like at vae.py:17). */
int vae_randn(float* out, int64_t n, uint64_t* rng, vae_stream_t stream);

/* Batched slidingWindow.__getitem__ (train.py:21-32) on a series resident in HBM:
 * out[b][c][l] = series[c*T + starts[b] + l] for b < B, c < C, l < L; starts: B device int64 window indices.
 * A start outside [0, T-L] (the reference returns None there) yields NaNs.  Replaces the DataLoader collate +
 * host-to-device copy in front of train.py:57. */
int vae_window_gather(const float* series, int C, int64_t T, int L, const int64_t* starts, int B, float* out,
                      vae_stream_t stream);

/* hipGraph capture of an arbitrary sequence of the calls above: begin, issue calls on the
 * returned stream, end -> handle; launch replays on `stream`. */
int vae_graph_begin(vae_stream_t* capture_stream);
int vae_graph_end(void** graph_handle);
int vae_graph_launch(void* graph_handle, vae_stream_t stream);
int vae_graph_destroy(void* graph_handle);

#ifdef __cplusplus
}
#endif
#endif
