"""Loss trajectory of the fused train step against the CPU oracle's, same weights / data / noise (diagnostic).
usage: traj_vs_oracle.py [B] [steps]   (multiscale model, x = 50 + 10 randn as bench.py's synthetic windows)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vae_amd
from oracle import vae_oracle as O
from vae_amd.train import FusedTrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
torch.manual_seed(0)
m = vae_amd.multiscale_vae.vae(3, 60, 15)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
gen = torch.Generator().manual_seed(1)
x = 50 + 10 * torch.randn(B, 3, 60, generator=gen)
m = m.cuda().train()
st = FusedTrainStep(m, vae_amd.optim.Adam(m.parameters(), lr=1e-3), B, "cuda", use_graph=True)
st.inject_eps = True
sd2 = {k: v.clone() for k, v in sd.items()}
opt = O.Adam(sd2, lr=1e-3)
for i in range(steps):
    eps = torch.randn(B, 30, generator=gen)
    st.eps.copy_(eps)
    l_hip = float(st.step(x.cuda()).item())
    ref = O.loss_and_grads("multiscale", sd2, x, eps)
    l_ref = float(ref["loss"])
    O.train_step("multiscale", sd2, opt, x, eps)
    print(f"step {i:2d}  hip {l_hip / B:14.6g}  oracle {l_ref / B:14.6g}  rel {abs(l_hip - l_ref) / max(abs(l_ref), 1e-30):.2e}", flush=True)
