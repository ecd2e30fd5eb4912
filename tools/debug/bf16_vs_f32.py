"""Gradient difference between the bf16 and the fp32 compute dtype on the same inputs, per tensor (relative L2)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import vae_amd
from vae_amd.train import FusedTrainStep
B = int(os.environ.get("B", 4096)); name = os.environ.get("MODEL", "mnist")
def run(dtype):
    vae_amd.set_compute_dtype(dtype)
    torch.manual_seed(0)
    if name == "mnist":
        m = vae_amd.mnist_vae.VariationalAutoencoder(15)
        x = torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(1))
    elif name == "wide":
        m = vae_amd.mnist_vae.VariationalAutoencoder(256, channels=(32, 64, 128), hidden=100)
        x = torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(1))
    else:
        m = vae_amd.multiscale_vae.vae(3, 60, 15)
        x = 50 + 10 * torch.randn(B, 3, 60, generator=torch.Generator().manual_seed(1))
    m = m.cuda().train()
    opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
    st = FusedTrainStep(m, opt, B, "cuda", use_graph=False, allow_seq1d=False)
    st.inject_eps = True
    st.eps.copy_(torch.randn(B, m.latent_total, generator=torch.Generator().manual_seed(2)))
    st.step(x.cuda()); torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters()}, float(st.loss.item()) / B
g0, l0 = run("f32"); g1, l1 = run("bf16")
print("loss/sample", l0, l1)
for k in g0:
    d = (g1[k] - g0[k]).double().norm() / g0[k].double().norm().clamp_min(1e-30)
    print(f"{k:50s} {float(d):.2e}   |g| {float(g0[k].norm()):.2e}")
