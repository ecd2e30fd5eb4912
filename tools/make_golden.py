#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference on disk); the GPU box never
sees the reference, only the .npz files this script wrote.  Nothing from the reference
is copied: the script imports its modules (Encoders/Decoders/vae/multiscale_vae/train)
and, for the MNIST Conv2d VAE that exists only as notebook source, execs notebook
cells 0-4 of .ipynb_checkpoints/vae-checkpoint.ipynb (cells 5+ download MNIST and are
never executed).  torchvision is not installed, its two unused symbols are stubbed.

Each case stores: state_dict before training, inputs x, injected eps per step, forward
outputs, ELBO terms, gradients of step 0, BatchNorm running stats after step 0, the
full state_dict after K=3 torch.optim.Adam steps and an eval-mode forward after them.

eps injection: the reference draws eps with torch.randn_like inside
reparametrization_trick (vae.py:15-18, multiscale_vae.py:206-209); we swap
torch.randn_like for a function returning the recorded eps while the reference's own
forward() runs, so the vectors pin the reference's real forward, not a re-derivation.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tvt.ToTensor = object
tvd = types.ModuleType("torchvision.datasets")
tvd.MNIST = object
sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.datasets": tvd})
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import Decoders  # noqa: E402
import Encoders  # noqa: E402
import multiscale_vae  # noqa: E402
import train as ref_train  # noqa: E402
import vae as ref_vae  # noqa: E402

K_STEPS = 3
LR = 1e-3


def load_mnist_notebook():
    nb = json.load(open(os.path.join(REF, ".ipynb_checkpoints", "vae-checkpoint.ipynb")))
    ns = {}
    for cell in nb["cells"][:5]:
        exec("".join(cell["source"]), ns)
    return ns


class inject_eps:
    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        self.saved = torch.randn_like
        torch.randn_like = lambda t, *a, **k: self.eps.to(t.dtype)

    def __exit__(self, *a):
        torch.randn_like = self.saved


def run_case(name, model, criterion, x, target_fn, latent_total):
    rec = {"x": x.numpy()}
    for k, v in model.state_dict().items():
        rec["sd/" + k] = v.detach().numpy().copy()
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    model.train()
    B = x.shape[0]
    for step in range(K_STEPS):
        eps = torch.randn(B, latent_total)
        rec[f"eps{step}"] = eps.numpy()
        opt.zero_grad()
        with inject_eps(eps):
            x_rec, mu, logvar = model(x)
        loss = criterion(x_rec, target_fn(x), mu, logvar)
        loss.backward()
        rec[f"loss{step}"] = np.float32(loss.item())
        if step == 0:
            rec["out/x_rec"] = x_rec.detach().numpy().copy()
            rec["out/mu"] = mu.detach().numpy().copy()
            rec["out/logvar"] = logvar.detach().numpy().copy()
            rec["out/z"] = (mu + eps * torch.exp(0.5 * logvar)).detach().numpy().copy()
            kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
            rec["out/kld"] = np.float32(kld.item())
            rec["out/recon"] = np.float32(loss.item() - kld.item())
            for k, p in model.named_parameters():
                rec["grad/" + k] = p.grad.detach().numpy().copy()
        opt.step()
        if step == 0:
            for k, v in model.state_dict().items():
                if "running_" in k or "num_batches" in k:
                    rec["sd1/" + k] = v.detach().numpy().copy()
    for k, v in model.state_dict().items():
        rec[f"sd{K_STEPS}/" + k] = v.detach().numpy().copy()
    model.eval()
    eps = torch.randn(B, latent_total)
    rec["eps_eval"] = eps.numpy()
    with torch.no_grad(), inject_eps(eps):
        x_rec, mu, logvar = model(x)
        loss = criterion(x_rec, target_fn(x), mu, logvar)
    rec["eval/x_rec"] = x_rec.numpy().copy()
    rec["eval/mu"] = mu.numpy().copy()
    rec["eval/logvar"] = logvar.numpy().copy()
    rec["eval/loss"] = np.float32(loss.item())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print(name, "loss0=%.4f" % rec["loss0"], "->", os.path.relpath(path), os.path.getsize(path), "bytes")


def series_windows(B, C, L, gen):
    # dataGen-shaped windows: per-channel level + sinusoid + unit noise
    t = torch.arange(L + B, dtype=torch.float32)
    level = torch.randint(0, 100, (C, 1), generator=gen).float()
    amp = 5 + 15 * torch.rand(C, 1, generator=gen)
    series = level + amp * torch.sin(2 * np.pi * t / 288.0) + torch.randn(C, L + B, generator=gen)
    return torch.stack([series[:, i:i + L] for i in range(B)]).contiguous()


def main():
    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(1234)

    def tcvae(C, N, lat, L, slope, fk):
        return ref_vae.VariationalAutoencoder(
            input_size=C, hidden_size=30, num_layers=N, latent_dims=lat,
            v_encoder=Encoders.LongShort_TCVAE_Encoder, v_decoder=Decoders.LongShort_TCVAE_Decoder,
            L=L, slope=slope, first_kernel=fk)

    first_col = lambda x: x[:, :, 0]
    # T: draft_pipe.ipynb config
    torch.manual_seed(11)
    run_case("tcvae_c3_n3_l60_fk21", tcvae(3, 3, 15, 60, 0.2, 21), ref_train.criterion,
             series_windows(8, 3, 60, g), first_col, 30)
    # T4: draft_pipe-checkpoint config (4 layers, first_kernel 15)
    torch.manual_seed(12)
    run_case("tcvae_c3_n4_l60_fk15", tcvae(3, 4, 15, 60, 0.2, 15), ref_train.criterion,
             series_windows(6, 3, 60, g), first_col, 30)
    # slope 0 (what train.objective always samples), first_kernel None, default L=30
    torch.manual_seed(13)
    run_case("tcvae_c2_n2_l30_relu", tcvae(2, 2, 4, 30, 0.0, None), ref_train.criterion,
             series_windows(5, 2, 30, g), first_col, 8)
    # S: multiscale_vae.vae(3, 60, 15)
    torch.manual_seed(14)
    run_case("multiscale_c3_l60", multiscale_vae.vae(3, 60, 15), ref_train.criterion,
             series_windows(8, 3, 60, g), first_col, 30)
    # M: notebook Conv2d VAE, BCE ELBO
    ns = load_mnist_notebook()
    torch.manual_seed(15)
    run_case("mnist_lat15", ns["VariationalAutoencoder"](15), ns["criterion"],
             torch.rand(8, 1, 28, 28, generator=g), lambda x: x, 15)


if __name__ == "__main__":
    main()
