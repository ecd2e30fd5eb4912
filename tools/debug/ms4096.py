import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_golden, sub, truth64
from test_cpu_host import _build
from oracle import vae_oracle as O
import vae_amd
from vae_amd.train import FusedTrainStep
B = int(os.environ.get("B", 4096))
stem, family = "multiscale_c3_l60", "multiscale"
g = load_golden(stem); sd = sub(g, "sd/")
gen = torch.Generator().manual_seed(11)
x = 50 + 10 * torch.randn(B, 3, 60, generator=gen); eps = torch.randn(B, 30, generator=gen)
ref = O.loss_and_grads(family, sd, x, eps)
tr = truth64(family, sd, x, eps)
m = _build(stem); m.load_state_dict(sd); m.to("cuda").train()
opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
st = FusedTrainStep(m, opt, B, "cuda", use_graph=False); st.inject_eps = True; st.eps.copy_(eps)
st.step(x.cuda())
gmax = max(float(v.abs().max()) for v in ref["grads"].values())
print("gmax", gmax)
for k, p in m.named_parameters():
    a = p.grad.cpu().double(); r = ref["grads"][k].double(); t = tr[k].double()
    print(f"{k:40s} |t|max {float(t.abs().max()):10.3e} ours-truth {float((a-t).abs().max()):9.3e} ref-truth {float((r-t).abs().max()):9.3e}")
k = "lencoder.encoder_cnn.0.weight"
a = m.get_parameter(k).grad.cpu().double(); t = tr[k].double()
print(((a - t)).reshape(6, 45))
