"""HipModule: nn.Module base whose forward/backward run as one libvae_amd program each.

Parameters stay ordinary nn.Parameters under the reference's state_dict keys, but their
storage is packed into one flat fp32 buffer (so the whole model is one PARAM address space
for the kernels, one bucket for the gradient all-reduce and one launch for Adam).  BatchNorm
running statistics are packed the same way into a STATE buffer.
"""
import torch
import torch.nn as nn

from . import _lib as L
from .program import ProgramBuilder

PARAM_ALIGN = 4  # floats


# ----------------------------------------------------------------------------- param holders
class ConvParams(nn.Module):
    """weight/bias of a Conv or ConvTranspose layer, initialised like torch.nn's own layers
    (kaiming_uniform(a=sqrt(5)) => U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for both)."""

    def __init__(self, weight_shape, bias_len, fan_in):
        super().__init__()
        bound = 1.0 / fan_in ** 0.5
        self.weight = nn.Parameter(torch.empty(weight_shape).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(bias_len).uniform_(-bound, bound))


class LinearParams(ConvParams):
    def __init__(self, in_features, out_features):
        super().__init__((out_features, in_features), out_features, in_features)


class NoParams(nn.Module):
    """Placeholder that keeps the reference's module indices (its activation modules)."""


def _bn_flush_hook(module, prefix, keep_vars):
    module.flush()


class BatchNormParams(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(C))
        self.bias = nn.Parameter(torch.zeros(C))
        self.register_buffer("running_mean", torch.zeros(C))
        self.register_buffer("running_var", torch.ones(C))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending = 0  # train-mode forwards not yet added to num_batches_tracked
        self.register_state_dict_pre_hook(_bn_flush_hook)

    def flush(self):
        if self._pending:
            self.num_batches_tracked += self._pending
            self._pending = 0


# ------------------------------------------------------------------------------- flat packing
def _pack(tensors, device):
    """Returns (flat, offsets, repacked).  If the tensors already sit back to back (aligned)
    in one storage on `device`, returns a zero-copy flat view; otherwise packs them."""
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += (t.numel() + PARAM_ALIGN - 1) // PARAM_ALIGN * PARAM_ALIGN
    if not tensors:
        return torch.empty(0, device=device), offs, False
    first = tensors[0]
    ok = first.device == device
    if ok:
        base, store = first.data_ptr(), first.untyped_storage().data_ptr()
        for t, o in zip(tensors, offs):
            if (t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != base + 4 * o
                    or t.untyped_storage().data_ptr() != store):
                ok = False
                break
        ok = ok and first.storage_offset() + total <= first.untyped_storage().nbytes() // 4
    if ok:
        return torch.as_strided(first.detach(), (total,), (1,), first.storage_offset()), offs, False
    flat = torch.zeros(total, dtype=torch.float32, device=device)
    for t, o in zip(tensors, offs):
        if t.dtype != torch.float32:
            raise L.VaeAmdError(f"vae_amd modules are fp32 only (got {t.dtype})")
        flat[o:o + t.numel()].copy_(t.detach().reshape(-1))
        t.data = flat[o:o + t.numel()].view(t.shape)
    return flat, offs, True


class HipModule(nn.Module):
    """Subclasses implement emit_program(b, training) that declares inputs/outputs on the
    ProgramBuilder `b` and returns nothing; forward() then goes through run_program()."""

    _TRANSIENT = ("_programs", "_flat", "_sflat", "_param_list", "_param_off", "_state_off", "_stepper", "_rng")

    def __getstate__(self):  # torch.save(model) / copy.deepcopy: drop ctypes programs and flat views
        d = self.__dict__.copy()
        for k in self._TRANSIENT:
            d.pop(k, None)
        return d

    def _float_buffers(self):
        return [b for _, b in self.named_buffers() if b.dtype.is_floating_point]

    def _bn_holders(self):
        return [m for m in self.modules() if isinstance(m, BatchNormParams)]

    def pack(self, device):
        """(Re)packs parameters and BN statistics; cheap no-op check when already packed."""
        params = [p for _, p in self.named_parameters()]
        flat, offs, _ = _pack(params, device)
        sflat, soffs, _ = _pack(self._float_buffers(), device)
        self._flat, self._param_list = flat, params
        self._param_off = {id(p): o for p, o in zip(params, offs)}
        self._sflat = sflat
        self._state_off = {id(b): o for b, o in zip(self._float_buffers(), soffs)}
        return flat, sflat

    def compiled(self, B, training, key_extra=()):
        key = (B, bool(training), torch.is_grad_enabled(), L.get_compute_dtype()) + tuple(key_extra)
        cache = self.__dict__.setdefault("_programs", {})
        sig = (tuple(self._param_off.values()), tuple(self._state_off.values()))
        if cache.get("sig") != sig:
            cache.clear()
            cache["sig"] = sig
        if key not in cache:
            b = ProgramBuilder(B, self._param_off, self._state_off, want_grads=torch.is_grad_enabled(),
                               nflat=self._flat.numel())
            self.emit_program(b, training)
            cache[key] = b.build()
        return cache[key]

    def run_program(self, inputs):
        """inputs: list of tensors in the order emit_program declares them.  Returns a dict of
        output tensors (views of a per-call workspace, so they stay valid like fresh tensors)."""
        x0 = inputs[0]
        L.require_device(x0)
        B = x0.shape[0]
        flat, sflat = self.pack(x0.device)
        prog = self.compiled(B, self.training)
        ext = [t.detach().to(torch.float32).contiguous() for t in inputs]
        if self.training:
            for m in self._bn_holders():
                m._pending += 1
        names = list(prog.outputs)
        if not torch.is_grad_enabled():
            work = prog.new_workspace(x0.device)
            prog.forward(work, flat, sflat, ext, self.training)
            return {n: prog.view(work, *prog.outputs[n]) for n in names}
        if not self.training and self._bn_holders():
            # The backward kernels implement the train-mode BatchNorm gradient (batch-statistic terms included);
            # eval-mode BatchNorm has the affine-only gradient, which is not built.  model.eval(); model(x) without
            # torch.no_grad() is legal and common (the reference itself only runs eval under no_grad, train.py:77-79),
            # so the forward runs (forward-only program) and only a backward() THROUGH its outputs raises.
            with torch.no_grad():
                prog = self.compiled(B, False)
                work = prog.new_workspace(x0.device)
                prog.forward(work, flat, sflat, ext, False)
                outs = [prog.view(work, *prog.outputs[n]) for n in prog.outputs]
            if any(p.requires_grad for p in self._param_list):
                outs = _EvalOutputsFn.apply(self._param_list[0], *outs)
            return dict(zip(list(prog.outputs), outs))
        outs = _ProgramFn.apply(self, prog, len(inputs), *inputs, *self._param_list)
        return dict(zip(names, outs))


def storage_alias(t):
    """Second tensor over t's memory with its OWN autograd version counter.  Program outputs are
    views of the per-call workspace; backward writes gradient regions of that same workspace, and
    doing so through the original tensor would bump the version the output views share."""
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.shape,
                                                                  t.stride())


class _EvalOutputsFn(torch.autograd.Function):
    """Marks eval-mode outputs as differentiable in name only: backward through them is refused (see run_program)."""

    @staticmethod
    def forward(ctx, anchor, *outs):
        return tuple(o.view_as(o) for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        raise L.VaeAmdError("vae_amd: gradients through an eval-mode BatchNorm are not supported; call the model "
                            "under torch.no_grad() in eval mode (as train.py:test does) or switch to .train()")


class _ProgramFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, prog, n_in, *args):
        inputs = args[:n_in]
        ext = [t.detach().to(torch.float32).contiguous() for t in inputs]
        work = prog.new_workspace(ext[0].device)
        prog.forward(work, module._flat, module._sflat, ext, module.training)
        ctx.module, ctx.prog, ctx.work, ctx.ext, ctx.n_in = module, prog, work, ext, n_in
        ctx.flat = module._flat
        ctx.shapes = [(module._param_off[id(p)], p.shape) for p in module._param_list]
        ctx.in_shapes = [t.shape for t in inputs]
        return tuple(prog.view(work, *prog.outputs[n]) for n in prog.outputs)

    @staticmethod
    def backward(ctx, *gouts):
        prog, work = ctx.prog, storage_alias(ctx.work)
        for (name, (buf, shape)), g in zip(prog.outputs.items(), gouts):
            gv = prog.grad_view(work, buf, shape)
            if g is None:
                gv.zero_()
            else:
                gv.copy_(g)
        grads = torch.empty_like(ctx.flat)
        prog.backward(work, ctx.flat, grads, ctx.module._sflat, ctx.ext)
        gin = []
        for (name, per_sample, buf), shp, need in zip(prog.inputs, ctx.in_shapes, ctx.needs_input_grad[3:3 + ctx.n_in]):
            gin.append(prog.grad_view(work, buf).view(shp) if (need and buf.gref >= 0) else None)
        gparams = [grads[o:o + shape.numel()].view(shape) for o, shape in ctx.shapes]
        return (None, None, None, *gin, *gparams)
