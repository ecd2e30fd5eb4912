import os, sys, torch
sys.path.insert(0, os.getcwd())
import vae_amd
from vae_amd.train import FusedTrainStep
B = int(os.environ.get("B", 4096)); model = os.environ.get("MODEL", "multiscale")
torch.manual_seed(0)
m = (vae_amd.mnist_vae.VariationalAutoencoder(15) if model == "mnist" else vae_amd.multiscale_vae.vae(3, 60, 15)).cuda().train()
opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
st = FusedTrainStep(m, opt, B, "cuda", use_graph=False)
x = torch.rand(B, 1, 28, 28, device="cuda") if model == "mnist" else 50 + 10 * torch.randn(B, 3, 60, device="cuda")
for i in range(int(os.environ.get("STEPS", 3))):
    st.step(x); torch.cuda.synchronize(); print("---- step", i, flush=True)
