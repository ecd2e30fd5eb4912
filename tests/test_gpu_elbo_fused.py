"""elbo_fused_k (elem.h; a train step's loss terms and d(loss)/d(x_rec) in one pass, the sum of the workgroups' pairs on
the weight-gradient reduction) against the two-launch form it replaces (elbo_fwd_k + elbo_bwd_k, VAE_AMD_ELBO_FUSED=0):
the grid and the loop are elbo_fwd_k's, so loss, recon, kld and the parameters after three Adam steps must be equal
BIT FOR BIT, for the BCE criterion (MNIST model, train.py:39-48 / notebook :141-151) and at a batch that is not a
multiple of anything.  The switch is read once per process, hence two child processes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, hashlib, torch
sys.path.insert(0, %r)
import vae_amd
from vae_amd.train import FusedTrainStep
B = int(sys.argv[1])
torch.manual_seed(3)
m = vae_amd.mnist_vae.VariationalAutoencoder(15).cuda().train()
st = FusedTrainStep(m, vae_amd.optim.Adam(m.parameters(), lr=1e-3), B, 'cuda', use_graph=(sys.argv[2] == 'graph'))
st.inject_eps = True
g = torch.Generator(device='cuda').manual_seed(4)
out = []
for step in range(3):
    x = torch.rand(B, 1, 28, 28, device='cuda', generator=g)
    st.eps.copy_(torch.randn(B, 15, device='cuda', generator=g))
    st.step(x)
    torch.cuda.synchronize()
    out.append(st.prog.scalar_view(st.work, st.out_ref, 4).cpu().numpy().tobytes().hex())  # loss, recon, kld, running total
print('TERMS', ' '.join(out))
print('PARAMS', hashlib.sha256(st.flat.cpu().numpy().tobytes()).hexdigest())
""" % ROOT


def _run(fused, B, mode):
    env = dict(os.environ, VAE_AMD_ELBO_FUSED="1" if fused else "0")
    out = subprocess.run([sys.executable, "-c", CODE, str(B), mode], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = {l.split()[0]: l for l in out.stdout.splitlines() if l.startswith(("TERMS", "PARAMS"))}
    assert set(lines) == {"TERMS", "PARAMS"}, out.stdout[-1500:]
    return lines


@pytest.mark.parametrize("B,mode", [(4096, "graph"), (301, "eager")])
def test_fused_elbo_equals_the_two_launch_form_bit_for_bit(B, mode):
    a, b = _run(True, B, mode), _run(False, B, mode)
    assert a["TERMS"] == b["TERMS"], (a["TERMS"], b["TERMS"])
    assert a["PARAMS"] == b["PARAMS"]
