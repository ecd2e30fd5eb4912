"""Program builder and runner: the host side of libvae_amd's op-program executor.

A model's forward is lowered ONCE per (batch size, mode) into an array of `vae_op` records
(include/vae_amd.h) plus a workspace layout; running it is a single C call for the whole
forward and one for the whole backward, so Python is not on the per-layer path.

Address spaces: 0 WORK (activations, activation grads, scratch), 1 PARAM (flat parameters),
2 GRAD (flat parameter grads), 3 STATE (BatchNorm running statistics), 4.. one per program input.
"""
import ctypes
import os

import torch

from . import _lib as L

ALIGN = 64  # floats; every workspace buffer starts on a 256-byte boundary


def mkref(space, off):
    return (space << 56) | off


class Buf:
    """A [B][ld] fp32 matrix view inside an address space (ld = floats per sample)."""

    __slots__ = ("ref", "gref", "ld", "n")

    def __init__(self, ref, gref, ld, n):
        self.ref, self.gref, self.ld, self.n = ref, gref, ld, n

    def cols(self, off, n):
        assert 0 <= off and off + n <= self.n
        return Buf(self.ref + off, self.gref + off if self.gref >= 0 else -1, self.ld, n)


class ProgramBuilder:
    def __init__(self, B, param_off, state_off, want_grads=True, nflat=0, allow_seq1d=True):
        """param_off / state_off: dict id(tensor) -> float offset in the flat PARAM / STATE buffers.
        nflat: floats of the flat PARAM buffer (needed for the whole-model 1-D path, include/vae_amd.h
        vae_seq_scratch_floats); allow_seq1d=False keeps a program on the per-layer path (data parallel)."""
        self.B = B
        self.nflat, self.allow_seq1d = int(nflat), allow_seq1d
        self.ops = []
        self.work_size = 0
        self.param_off, self.state_off = param_off, state_off
        self.want_grads = want_grads
        self.inputs = []  # (name, per_sample, Buf)
        self.outputs = {}  # name -> (Buf, shape_per_sample)
        self.scratch_need = 2 * L.ELBO_BLOCKS
        self._scratch_ops = []  # (op, slot, floats)

    # -- allocation ---------------------------------------------------------------------
    def _alloc(self, floats):
        off = self.work_size
        self.work_size += (floats + ALIGN - 1) // ALIGN * ALIGN
        return off

    def act(self, per_sample, grad=True):
        ref = mkref(L.SP_WORK, self._alloc(per_sample * self.B))
        gref = mkref(L.SP_WORK, self._alloc(per_sample * self.B)) if (grad and self.want_grads) else -1
        return Buf(ref, gref, per_sample, per_sample)

    def fixed(self, floats):
        return mkref(L.SP_WORK, self._alloc(floats))

    def input(self, name, per_sample, grad=False):
        space = L.SP_EXT0 + len(self.inputs)
        gref = mkref(L.SP_WORK, self._alloc(per_sample * self.B)) if (grad and self.want_grads) else -1
        buf = Buf(mkref(space, 0), gref, per_sample, per_sample)
        self.inputs.append((name, per_sample, buf))
        return buf

    def output(self, name, buf, shape):
        self.outputs[name] = (buf, tuple(shape))

    def param(self, t):
        return mkref(L.SP_PARAM, self.param_off[id(t)])

    def state(self, t):
        return mkref(L.SP_STATE, self.state_off[id(t)])

    def _op(self, kind, act=L.ACT_NONE, flags=0, r=(), d=(), f=()):
        op = L.VaeOp()
        op.kind, op.act, op.flags = kind, act, flags
        for i in range(12):
            op.r[i] = -1
        for i, v in enumerate(r):
            op.r[i] = v
        for i, v in enumerate(d):
            op.d[i] = v
        for i, v in enumerate(f):
            op.f[i] = v
        self.ops.append(op)
        return op

    def _need_scratch(self, op, slot, floats):
        self.scratch_need = max(self.scratch_need, floats)
        self._scratch_ops.append((op, slot))

    # -- ops ------------------------------------------------------------------------------
    def conv(self, x, w, b, cin, hw, cout, k, stride, pad, transposed=False, out_pad=(0, 0), act=L.ACT_NONE,
             slope=0.0, need_din=True):
        """x: Buf of cin*H*W per sample.  hw/k/stride/pad/out_pad are (h, w) pairs.  Returns (y, (OH, OW))."""
        (H, W), (KH, KW), (sh, sw), (ph, pw) = hw, k, stride, pad
        if transposed:
            OH = (H - 1) * sh - 2 * ph + KH + out_pad[0]
            OW = (W - 1) * sw - 2 * pw + KW + out_pad[1]
        else:
            OH, OW = (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1
        if OH < 1 or OW < 1:
            raise ValueError("Window Length is too small in relation to the number of Layers")
        assert x.n == cin * H * W and x.ld == x.n, "conv input must be a dense [B][C*H*W] buffer"
        y = self.act(cout * OH * OW)
        op = self._op(L.OP_CONVT if transposed else L.OP_CONV, act=act,
                      r=(x.ref, y.ref, x.gref if need_din else -1, y.gref, self.param(w),
                         self.param(b) if b is not None else -1),
                      d=(cin, H, W, cout, OH, OW, KH, KW, sh, sw, ph, pw), f=(slope,))
        self._need_scratch(op, 6, 2 * cout * L.MAX_CHUNKS)
        return y, (OH, OW)

    def batchnorm(self, x, bn, C, HW, relu=False, eps=1e-5, momentum=0.1):
        """bn: holder with .weight .bias .running_mean .running_var"""
        assert x.n == C * HW and x.ld == x.n
        y = self.act(C * HW)
        save = -1  # allocated in build(): the save blocks (+ accumulators) of all BatchNorms lie back to back
        op = self._op(L.OP_BN, flags=L.F_BN_RELU if relu else 0,
                      r=(x.ref, y.ref, x.gref, y.gref, self.param(bn.weight), self.param(bn.bias), save,
                         self.state(bn.running_mean), self.state(bn.running_var)),
                      d=(C, HW), f=(eps, momentum))
        self._need_scratch(op, 9, 2 * C * L.MAX_CHUNKS + 4 * C * L.MAX_WORLD)  # partials + data-parallel rows
        return y

    def linear(self, x, w, b, K, N, out=None, act=L.ACT_NONE, acc_din=False, need_din=True):
        assert x.n == K, (x.n, K)
        y = out if out is not None else self.act(N)
        assert y.n == N
        op = self._op(L.OP_LINEAR, act=act, flags=L.F_ACC_DIN if acc_din else 0,
                      r=(x.ref, y.ref, x.gref if need_din else -1, y.gref, self.param(w),
                         self.param(b) if b is not None else -1),
                      d=(K, N, x.ld, y.ld, x.ld, y.ld), f=(0.0,))
        self._need_scratch(op, 6, 2 * N * L.MAX_CHUNKS)
        if self.want_grads:  # per-slice rows of the weight gradient (summed in a fixed order: deterministic)
            op.r[10] = self.fixed(L.lib().vae_linear_wgrad_partial_floats(ctypes.byref(op), self.B))
        return y

    def reparam(self, mu, logvar, eps, latent, ext_grad=None, kl_grad=False, gscale=-1):
        """ext_grad: (g_mu Buf, g_logvar Buf) incoming gradients added in backward (autograd path)."""
        assert mu.ld == latent and logvar.ld == latent and eps.ld == latent
        z = self.act(latent)
        flags = (L.F_EXT_GRAD if ext_grad else 0) | (L.F_KL_GRAD if kl_grad else 0)
        self._op(L.OP_REPARAM, flags=flags,
                 r=(mu.ref, logvar.ref, eps.ref, z.ref, z.gref, mu.gref, logvar.gref,
                    ext_grad[0].ref if ext_grad else -1, ext_grad[1].ref if ext_grad else -1, gscale),
                 d=(latent,))
        return z

    def slice0(self, x, C, Lw):
        assert x.n == C * Lw
        y = self.act(C)
        self._op(L.OP_SLICE0, r=(x.ref, y.ref, x.gref, y.gref), d=(C, Lw))
        return y

    def elbo(self, x_rec, target_ref, t_sb, t_se, mu, logvar, latent, bce, accumulate=True, loss_in_bwd=False):
        """loss_in_bwd: the sum of the partials into out[0..3] is left to the op's backward launch (train steps, which
        always run it); a forward-only run of such a program leaves `out` stale."""
        out = self.fixed(4)
        op = self._op(L.OP_ELBO, flags=(L.F_BCE if bce else 0) | (L.F_LOSS_IN_BWD if loss_in_bwd else 0),
                      r=(x_rec.ref, target_ref, mu.ref, logvar.ref, out, x_rec.gref, -1, -1),
                      d=(x_rec.n, t_sb, t_se, latent, 0, 1 if accumulate else 0))
        if loss_in_bwd:
            # the workgroups' partial pairs outlive the launch that writes them (elbo_fused_k at the head of the backward
            # pass; summed by the weight-gradient reduction at its end): a region of their own, not the shared scratch
            op.r[6] = self.fixed(2 * L.ELBO_BLOCKS)
        else:
            self._need_scratch(op, 6, 2 * L.ELBO_BLOCKS)
        return out

    # -- finish ---------------------------------------------------------------------------
    def _fuse(self):
        """Graph rewrite: Conv/ConvT ops that fit the LDS-staged MFMA kernels get VAE_F_FUSED, and a
        BatchNorm that sits between two such ops becomes virtual (never materialised): the producer's
        epilogue takes its statistics, the consumer normalises (+ReLU) on load, and the backward pass
        is folded into the same kernels (include/vae_amd.h, VAE_F_FUSED)."""
        lib = L.lib()
        convs = (L.OP_CONV, L.OP_CONVT)
        for op in self.ops:
            if op.kind in convs and lib.vae_conv_fused_ok(ctypes.byref(op), self.B):
                op.flags |= L.F_FUSED
                self.scratch_need = max(self.scratch_need, lib.vae_conv_scratch_floats(ctypes.byref(op), self.B))
                if self.want_grads:  # the layer's own weight-gradient partials (reduced in one launch at the end)
                    op.r[10] = self.fixed(lib.vae_conv_wgrad_partial_floats(ctypes.byref(op), self.B))
            elif op.kind in convs and self.want_grads:
                n = lib.vae_conv_wgrad_partial_floats(ctypes.byref(op), self.B)
                if n > 0:  # wide layer: split-reduction partials of its weight gradient
                    op.r[10] = self.fixed(n)
        out_refs = {buf.ref for buf, _ in self.outputs.values()}
        for bn in self.ops:
            if bn.kind != L.OP_BN or bn.r[1] in out_refs:
                continue
            prod = [o for o in self.ops if o.kind in convs and o.r[1] == bn.r[0]]
            cons = [o for o in self.ops if o is not bn and o.r[0] == bn.r[1]]
            if len(prod) != 1 or len(cons) != 1 or cons[0].kind not in convs:
                continue
            p, c = prod[0], cons[0]
            if not (p.flags & L.F_FUSED and c.flags & L.F_FUSED):
                continue
            if bn.flags & L.F_BN_ACC:
                p.flags |= L.F_BN_ACC
                c.flags |= L.F_BN_ACC
            c.r[0], c.r[2], c.r[7] = bn.r[0], bn.r[2], bn.r[6]
            if bn.flags & L.F_BN_RELU:
                c.flags |= L.F_IN_RELU
            p.r[8], p.r[9] = bn.r[6], bn.r[7]
            bn.flags |= L.F_FUSED
            bn.r[1] = bn.r[3] = -1

    def _seq1d(self):
        """A complete 1-D VAE program (TCVAE / multiscale) runs as one launch per BatchNorm-delimited phase:
        attach the scratch that path needs to ops[0].r[11].  Returns True when it applies."""
        if not (self.allow_seq1d and self.nflat > 0 and len(self.ops) >= 6):
            return False
        arr = (L.VaeOp * len(self.ops))(*self.ops)
        need = L.lib().vae_seq_scratch_floats(arr, len(self.ops), self.B, self.nflat)
        if need <= 0:
            return False
        self.ops[0].r[11] = self.fixed(need)
        return True

    def _bn_blocks(self):
        """Save blocks of the BatchNorm ops, back to back; blocks of <= 16 channels are followed by the fixed-point
        accumulators their statistics are summed in (include/vae_amd.h: VAE_F_BN_ACC), so that the executor clears them
        all with one memset and finalises such a BatchNorm in the prologue of its first consumer kernel."""
        lo = self.work_size
        for op in self.ops:
            if op.kind == L.OP_BN:
                C = op.d[0]
                acc = C <= L.BN_ACC_MAXC
                op.r[6] = self.fixed(8 * C + (L.bn_acc_floats(C) if acc else 0))
                if acc:
                    op.flags |= L.F_BN_ACC
        self.bn_arena = (lo, self.work_size - lo)  # floats of WORK: every save block and accumulator

    def build(self):
        self._bn_blocks()
        self.seq1d = self._seq1d()
        if not self.seq1d and os.environ.get("VAE_AMD_UNFUSED", "0") != "1":
            self._fuse()
        scratch = self.fixed(self.scratch_need)  # shared: ops run back to back on one stream
        for op, slot in self._scratch_ops:
            op.r[slot] = scratch
        arr = (L.VaeOp * len(self.ops))(*self.ops)
        prog = CompiledProgram(arr, len(self.ops), self.B, self.work_size, self.inputs, self.outputs)
        prog.seq1d = self.seq1d
        prog.bn_arena = self.bn_arena
        prog.ends_with_elbo = len(self.ops) > 0 and self.ops[-1].kind == L.OP_ELBO
        return prog


class CompiledProgram:
    def __init__(self, ops, n_ops, B, work_size, inputs, outputs):
        self.ops, self.n_ops, self.B, self.work_size = ops, n_ops, B, work_size
        self.inputs, self.outputs = inputs, outputs
        self.n_bases = L.SP_EXT0 + len(inputs)
        self._bases = (ctypes.c_void_p * self.n_bases)()
        self._sizes = (ctypes.c_int64 * self.n_bases)()
        # The workspace (by address) whose BatchNorm accumulators are known to be zero: a training forward over it may
        # then skip its memset (vae_run_forward training = 2).  Any forward dirties them; a whole backward pass of a
        # program that ends with the ELBO op clears them again (include/vae_amd.h).
        self._clean_ptr = None
        self.bn_arena, self.ends_with_elbo = (0, 0), False

    def ensure_clean(self, work):
        """Zeroes the BatchNorm accumulators of `work` unless they are known to be clean; afterwards they are."""
        if self._clean_ptr != work.data_ptr():
            lo, n = self.bn_arena
            if n:
                work[lo:lo + n].zero_()
            self._clean_ptr = work.data_ptr()

    def _after_backward(self, work, whole=True):
        self._clean_ptr = work.data_ptr() if (whole and self.ends_with_elbo) else None

    def new_workspace(self, device):
        return torch.empty(self.work_size, dtype=torch.float32, device=device)

    def _bind(self, work, params, grads, state, ext):
        assert work.numel() >= self.work_size and len(ext) == len(self.inputs)
        tensors = [work, params, grads, state] + list(ext)
        for i, t in enumerate(tensors):
            if t is None:
                self._bases[i], self._sizes[i] = None, 0
                continue
            assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda, "program tensors: contiguous fp32 on GPU"
            self._bases[i], self._sizes[i] = t.data_ptr(), t.numel()
        for (name, per_sample, _), t in zip(self.inputs, ext):
            if t.numel() != per_sample * self.B:
                raise ValueError(f"input '{name}': expected {self.B}x{per_sample} elements, got {tuple(t.shape)}")

    def forward(self, work, params, state, ext, training, stream=None):
        self._bind(work, params, None, state, ext)
        clean = training and self._clean_ptr == work.data_ptr()
        self._clean_ptr = None
        L.check(L.lib().vae_run_forward(self.ops, self.n_ops, self._bases, self._sizes, self.n_bases, self.B,
                                        (2 if clean else 1) if training else 0,
                                        stream if stream is not None else L.stream_ptr()),
                "vae_run_forward")

    def time_forward(self, work, params, state, ext, training, stream=None):
        """Forward with a HIP event between ops on the launch stream; returns per-op device ms (list)."""
        self._bind(work, params, None, state, ext)
        self._clean_ptr = None
        ms = (ctypes.c_float * self.n_ops)()
        L.check(L.lib().vae_time_forward(self.ops, self.n_ops, self._bases, self._sizes, self.n_bases, self.B,
                                         1 if training else 0, stream if stream is not None else L.stream_ptr(), ms),
                "vae_time_forward")
        return list(ms)

    def time_seq(self, work, params, grads, state, ext, stream=None):
        """Forward + backward of a fused 1-D program with a HIP event before every launch on the launch stream;
        returns [(tag, device ms, algorithmic bytes), ...] (vae_time_seq)."""
        self._bind(work, params, grads, state, ext)
        cap = 40
        ms, nbytes = (ctypes.c_float * cap)(), (ctypes.c_double * cap)()
        tags, n = ctypes.create_string_buffer(8 * cap), ctypes.c_int32(0)
        self._clean_ptr = None
        L.check(L.lib().vae_time_seq(self.ops, self.n_ops, self._bases, self._sizes, self.n_bases, self.B,
                                     stream if stream is not None else L.stream_ptr(), ms, nbytes, tags, cap,
                                     ctypes.byref(n)), "vae_time_seq")
        return [(tags.raw[8 * k:8 * k + 8].split(b"\0")[0].decode(), ms[k], nbytes[k]) for k in range(n.value)]

    def time_step(self, work, params, grads, state, ext, m=None, v=None, hyper=None, tick=None, stream=None, cap=256):
        """One REAL train step (forward, backward, Adam when hyper is given) with a HIP event behind every launch on
        the launch stream (vae_time_step); returns [(kernel name, device ms, algorithmic bytes, flops), ...]."""
        self._bind(work, params, grads, state, ext)
        ms, nbytes, flops = (ctypes.c_float * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
        names, n = ctypes.create_string_buffer(64 * cap), ctypes.c_int32(0)
        L.check(L.lib().vae_time_step(self.ops, self.n_ops, self._bases, self._sizes, self.n_bases, self.B,
                                      stream if stream is not None else L.stream_ptr(), L.ptr(m), L.ptr(v), L.ptr(hyper),
                                      L.ptr(tick), ms, nbytes, flops, names, cap, ctypes.byref(n)), "vae_time_step")
        self._after_backward(work)
        return [(names.raw[64 * k:64 * k + 64].split(b"\0")[0].decode(), ms[k], nbytes[k], flops[k]) for k in range(n.value)]

    def backward(self, work, params, grads, state, ext, stream=None):
        self._bind(work, params, grads, state, ext)
        L.check(L.lib().vae_run_backward(self.ops, self.n_ops, self._bases, self._sizes, self.n_bases, self.B,
                                         stream if stream is not None else L.stream_ptr()),
                "vae_run_backward")
        self._after_backward(work)

    def backward_step(self, work, params, grads, state, ext, m, v, hyper, tick, stream=None, next_eps=None, rng=None):
        """Backward pass + Adam update of `params` (vae_run_backward_step); hyper = device {lr, beta1, beta2, eps}.
        next_eps / rng: the Adam launch also draws the next step's noise into next_eps."""
        self._bind(work, params, grads, state, ext)
        L.check(L.lib().vae_run_backward_step(self.ops, self.n_ops, self._bases, self._sizes, self.n_bases, self.B,
                                              stream if stream is not None else L.stream_ptr(), L.ptr(m), L.ptr(v),
                                              L.ptr(hyper), L.ptr(tick), L.ptr(next_eps),
                                              next_eps.numel() if next_eps is not None else 0, L.ptr(rng)),
                "vae_run_backward_step")
        self._after_backward(work)

    def backward_range(self, work, params, grads, state, ext, begin, end, zero_grads, stream=None):
        """Backward over ops[begin:end) only (vae_run_backward_range): the data-parallel step's gradient buckets."""
        self._bind(work, params, grads, state, ext)
        L.check(L.lib().vae_run_backward_range(self.ops, self.n_ops, self._bases, self._sizes, self.n_bases, self.B,
                                               stream if stream is not None else L.stream_ptr(), begin, end,
                                               1 if zero_grads else 0),
                "vae_run_backward_range")
        self._after_backward(work, whole=bool(zero_grads) and end == self.n_ops)

    def grad_buckets(self):
        """Splits the program at its REPARAM op into (decoder + loss, encoder) ranges and returns
        [(op_begin, op_end, grad_lo, grad_hi), ...] in backward order, or None when the parameter ranges of the two
        halves interleave in the flat buffer (then the whole gradient is one bucket)."""
        ks = [i for i in range(self.n_ops) if self.ops[i].kind == L.OP_REPARAM]
        if len(ks) != 1 or getattr(self, "seq1d", False):  # (a fused 1-D program's backward runs whole)
            return None
        k = ks[0]

        def span(lo, hi):
            offs = []
            for i in range(lo, hi):
                op = self.ops[i]
                if op.kind in (L.OP_CONV, L.OP_CONVT, L.OP_LINEAR, L.OP_BN):
                    offs += [(r & ((1 << 56) - 1)) for r in (op.r[4], op.r[5]) if r >= 0]
            return (min(offs), max(offs)) if offs else None
        dec, enc = span(k + 1, self.n_ops), span(0, k + 1)
        if dec is None or enc is None or not (enc[1] < dec[0]):
            return None
        return [(k + 1, self.n_ops, dec[0], None), (0, k + 1, 0, dec[0])]

    def view(self, work, buf, shape=None):
        """Tensor view of a dense WORK buffer ([B][buf.n]); shape = per-sample shape."""
        assert (buf.ref >> 56) == L.SP_WORK and buf.ld == buf.n
        off = buf.ref & ((1 << 56) - 1)
        t = work[off:off + self.B * buf.n]
        return t.view(self.B, *(shape if shape is not None else (buf.n,)))

    def grad_view(self, work, buf, shape=None):
        assert buf.gref >= 0 and buf.ld == buf.n
        off = buf.gref & ((1 << 56) - 1)
        t = work[off:off + self.B * buf.n]
        return t.view(self.B, *(shape if shape is not None else (buf.n,)))

    def scalar_view(self, work, ref, n):
        off = ref & ((1 << 56) - 1)
        return work[off:off + n]
