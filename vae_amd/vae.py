"""MI355X drop-in for the reference's VariationalAutoencoder wrapper (vae.py:8-23) and the
shared machinery of all three VAE families (generic TCVAE, multiscale_vae, notebook MNIST)."""
import ctypes

import torch

from . import _lib as L
from .module import HipModule, storage_alias
from .program import Buf


def device_randn(shape, rng, device):
    """eps ~ N(0,1) from libvae_amd's Philox kernel (replaces torch.randn_like, vae.py:17)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    L.check(L.lib().vae_randn(L.ptr(out), out.numel(), L.ptr(rng), L.stream_ptr()), "vae_randn")
    return out


class _ReparamFn(torch.autograd.Function):
    """Standalone reparametrization_trick(mu, logvar) for callers outside forward()
    (utils.py:23,52,85 in the reference call it directly)."""

    @staticmethod
    def forward(ctx, prog, mu, logvar, eps):
        work = prog.new_workspace(mu.device)
        ext = [t.detach().float().contiguous() for t in (mu, logvar, eps)]
        empty = torch.empty(0, device=mu.device)
        prog.forward(work, empty, empty, ext, False)
        ctx.prog, ctx.work, ctx.ext = prog, work, ext
        return prog.view(work, *prog.outputs["z"])

    @staticmethod
    def backward(ctx, gz):
        prog, work = ctx.prog, storage_alias(ctx.work)
        prog.grad_view(work, prog.outputs["z"][0]).copy_(gz)
        g = torch.empty(4, device=gz.device)
        prog.backward(work, g, g, g[:0], ctx.ext)
        return None, prog.grad_view(work, prog.inputs[0][2]), prog.grad_view(work, prog.inputs[1][2]), None


class HipVAE(HipModule):
    """Common forward: encoder -> reparameterisation -> decoder, returning (x_rec, mu, logvar).

    Subclasses define: in_shape (per-sample input shape), latent_total, out_shape,
    emit_encoder(b, x, mu, logvar), emit_decoder(b, z) -> x_rec Buf, and the ELBO they train
    with (elbo_bce, target_strides) for the fused train step."""

    elbo_bce = False

    def _rng_state(self, device):
        rng = self.__dict__.get("_rng")
        if rng is None or rng.device != device:
            # seeded from torch's global generator so torch.manual_seed() controls eps
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            rng = torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)  # {seed, offset, scratch}
            self.__dict__["_rng"] = rng
        return rng

    def _numel(self, shape):
        n = 1
        for s in shape:
            n *= s
        return n

    def emit_program(self, b, training):
        x = b.input("x", self._numel(self.in_shape))
        eps = b.input("eps", self.latent_total)
        mu, logvar = b.act(self.latent_total), b.act(self.latent_total)
        self.emit_encoder(b, x, mu, logvar)
        ext = None
        if b.want_grads:  # incoming d(loss)/d(mu), d(loss)/d(logvar) are added in place
            ext = (Buf(mu.gref, -1, mu.ld, mu.n), Buf(logvar.gref, -1, logvar.ld, logvar.n))
        z = b.reparam(mu, logvar, eps, self.latent_total, ext_grad=ext)
        x_rec = self.emit_decoder(b, z)
        b.output("x_rec", x_rec, self.out_shape)
        b.output("mu", mu, (self.latent_total,))
        b.output("logvar", logvar, (self.latent_total,))

    def _reparam_program(self, B, latent):
        from .program import ProgramBuilder
        cache = self.__dict__.setdefault("_programs", {})
        key = ("reparam", B, latent, torch.is_grad_enabled())
        if key not in cache:
            b = ProgramBuilder(B, {}, {}, want_grads=torch.is_grad_enabled())
            mu, lv = b.input("mu", latent, grad=True), b.input("logvar", latent, grad=True)
            eps = b.input("eps", latent)
            b.output("z", b.reparam(mu, lv, eps, latent), (latent,))
            cache[key] = b.build()
        return cache[key]

    def reparametrization_trick(self, mu, logvar, eps=None):
        """z = mu + eps * exp(0.5 * logvar); eps drawn on the device unless injected."""
        L.require_device(mu)
        if eps is None:
            eps = device_randn(mu.shape, self._rng_state(mu.device), mu.device)
        # the program is built here: grad mode is off inside autograd.Function.forward
        return _ReparamFn.apply(self._reparam_program(mu.shape[0], mu.shape[1]), mu, logvar, eps)

    def forward(self, x, eps=None):
        L.require_device(x)
        if x.shape[1:] != tuple(self.in_shape):
            raise ValueError(f"expected input of shape (B, {', '.join(map(str, self.in_shape))}), got {tuple(x.shape)}")
        if eps is None:
            eps = device_randn((x.shape[0], self.latent_total), self._rng_state(x.device), x.device)
        out = self.run_program([x, eps])
        return out["x_rec"], out["mu"], out["logvar"]


class VariationalAutoencoder(HipVAE):
    """vae.py:8-23.  v_encoder / v_decoder are classes called positionally as
    (input_size, hidden_size, num_layers, latent_dims, L, slope, first_kernel); any pair with an
    emit(b, ...) method plugs in (LongShort_TCVAE_Encoder / LongShort_TCVAE_Decoder ship here)."""

    def __init__(self, input_size, hidden_size, num_layers, latent_dims, v_encoder, v_decoder, L=30, slope=0.2,
                 first_kernel=None):
        super().__init__()
        self.encoder = v_encoder(input_size, hidden_size, num_layers, latent_dims, L, slope, first_kernel)
        self.decoder = v_decoder(input_size, hidden_size, num_layers, latent_dims, L, slope, first_kernel)
        self.in_shape = (input_size, L)
        self.latent_total = getattr(self.encoder, "latent_total", latent_dims)
        self.out_shape = (input_size,)

    def target_strides(self):
        """ELBO target is data[:, :, 0] (train.py:63): batch stride C*L, element stride L."""
        C, Lw = self.in_shape
        return C * Lw, Lw

    def emit_encoder(self, b, x, mu, logvar):
        self.encoder.emit(b, x, mu, logvar)

    def emit_decoder(self, b, z):
        return self.decoder.emit(b, z)
