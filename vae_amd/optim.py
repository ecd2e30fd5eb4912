"""Fused Adam for vae_amd models (replaces torch.optim.Adam at draft_pipe.py:132,
train.py:113, notebook :249; same defaults and update rule, no weight decay / amsgrad).

When the parameters are packed in one flat buffer (HipModule.pack) and their .grad tensors
are views of one flat gradient buffer (what the program backward produces), a step is ONE
kernel over the flat buffers; otherwise it is one launch of the same kernel per parameter.
"""
import torch

from . import _lib as L


def _flat_view(tensors):
    """Zero-copy flat view over tensors that sit back to back (4-float aligned) in one storage,
    else None.  Mirrors module._pack's layout rule."""
    from .module import PARAM_ALIGN
    first = tensors[0]
    base, store, total = first.data_ptr(), first.untyped_storage().data_ptr(), 0
    for t in tensors:
        if (t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != base + 4 * total
                or t.untyped_storage().data_ptr() != store):
            return None
        total += (t.numel() + PARAM_ALIGN - 1) // PARAM_ALIGN * PARAM_ALIGN
    if first.storage_offset() + total > first.untyped_storage().nbytes() // 4:
        return None
    return torch.as_strided(first.detach(), (total,), (1,), first.storage_offset())


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def flat_state(self, group, flat_p):
        """(exp_avg, exp_avg_sq, tick) matching the flat parameter view; created on first use.
        tick = int32[4] device block: {step, pad, 1-b1^t, 1-b2^t} (see vae_adam_step)."""
        st = self.state.setdefault("flat%d" % id(group), {})
        if st.get("m") is None or st["m"].numel() != flat_p.numel() or st["m"].device != flat_p.device:
            st["m"] = torch.zeros_like(flat_p)
            st["v"] = torch.zeros_like(flat_p)
            st["tick"] = torch.zeros(4, dtype=torch.int32, device=flat_p.device)
        return st["m"], st["v"], st["tick"]

    def launch(self, group, p, g, m, v, tick, stream=None, hyper=None):
        """hyper: device float[4] {lr, beta1, beta2, eps} read by the kernel at run time (captured graphs)."""
        if hyper is not None:
            L.check(L.lib().vae_adam_step_hyper(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), L.ptr(hyper),
                                                L.ptr(tick), stream if stream is not None else L.stream_ptr()),
                    "vae_adam_step_hyper")
            return
        b1, b2 = group["betas"]
        L.check(L.lib().vae_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), group["lr"], b1, b2,
                                      group["eps"], L.ptr(tick), stream if stream is not None else L.stream_ptr()),
                "vae_adam_step")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            L.require_device(params[0])
            all_params = group["params"]
            flat_p = _flat_view(all_params) if len(params) == len(all_params) else None
            flat_g = _flat_view([p.grad for p in params]) if flat_p is not None else None
            if flat_p is not None and flat_g is not None and flat_g.numel() == flat_p.numel():
                m, v, tick = self.flat_state(group, flat_p)
                self.launch(group, flat_p, flat_g, m, v, tick)
                continue
            for p in params:  # unpacked parameters: same kernel, one launch each
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
                    st["tick"] = torch.zeros(4, dtype=torch.int32, device=p.device)
                if not p.is_contiguous():
                    raise L.VaeAmdError("vae_amd.optim.Adam needs contiguous parameters")
                self.launch(group, p.data, p.grad.contiguous(), st["m"], st["v"], st["tick"])
        return loss
