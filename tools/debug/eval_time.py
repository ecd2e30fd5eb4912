"""Eval-mode forward of the 1-D families at B = 4096 through the module API (model.eval(); with torch.no_grad(): model(x)):
ms per call with the decoder evaluated at time index 0 inside the heads phase (VAE_SEQ1D_PRUNE=1, default) and with the
full decoder phases (VAE_SEQ1D_PRUNE=0); the two results must agree."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vae_amd  # noqa: E402

B, dev = 4096, "cuda:0"
torch.manual_seed(0)
for name in ("multiscale", "tcvae"):
    if name == "multiscale":
        m = vae_amd.multiscale_vae.vae(3, 60, 15)
    else:
        m = vae_amd.VariationalAutoencoder(input_size=3, hidden_size=30, num_layers=3, latent_dims=15,
                                           v_encoder=vae_amd.Encoders.LongShort_TCVAE_Encoder,
                                           v_decoder=vae_amd.Decoders.LongShort_TCVAE_Decoder, L=60, slope=0.2, first_kernel=21)
    m.to(dev).train()
    x = 50 + 10 * torch.randn(B, 3, 60, device=dev)
    eps = torch.randn(B, 30, device=dev)
    for _ in range(3):  # move the running statistics off their initial values
        m(x, eps=eps)
    m.eval()
    outs = {}
    for prune in ("1", "0"):
        os.environ["VAE_SEQ1D_PRUNE"] = prune
        with torch.no_grad():
            for _ in range(20):
                out = m(x, eps=eps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 300
            for _ in range(n):
                out = m(x, eps=eps)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n * 1e3
        outs[prune] = [o.clone() for o in out]
        print(f"{name}: eval forward B={B} VAE_SEQ1D_PRUNE={prune}: {dt:.4f} ms/call")
    for a, b in zip(outs["1"], outs["0"]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
print("pruned == full")
