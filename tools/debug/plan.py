"""Prints the fused 1-D path's plan (chunk size, LDS per phase) for a model; needs no GPU."""
import os, sys
os.environ.setdefault("VAE_SEQ1D_VERBOSE", "2")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vae_amd
from vae_amd.program import ProgramBuilder

name = sys.argv[1] if len(sys.argv) > 1 else "multiscale"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
model = vae_amd.multiscale_vae.vae(3, 60, 15) if name == "multiscale" else vae_amd.vae.Variational_Autoencoder(1, 64, 10, 2) if name == "tcvae" else None
flat, _ = model.pack("cpu")
b = ProgramBuilder(B, model._param_off, model._state_off, want_grads=True, nflat=flat.numel())
x = b.input("x", model._numel(model.in_shape))
eps = b.input("eps", model.latent_total)
mu, logvar = b.act(model.latent_total), b.act(model.latent_total)
model.emit_encoder(b, x, mu, logvar)
z = b.reparam(mu, logvar, eps, model.latent_total, kl_grad=True)
xr = model.emit_decoder(b, z)
print("seq1d:", b._seq1d())
