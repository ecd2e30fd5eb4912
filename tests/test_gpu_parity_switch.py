"""GPU parity, third file: (i) a model WITHOUT activation kinks (LeakyReLU slope 1.0 = identity,
Encoders.py:42-44 takes the slope as an argument) at the benched batch, on both 1-D paths, which must pass
without a single element going through a tolerance hatch of conftest.assert_grads_close; (ii) the MNIST model
on both sides of the executor's path switches: the Linear/reparam run as one launch (fc_chain_k) or op by op
(VAE_AMD_FC, FC_AUTO_MAX_BATCH = 12,288 in vae_amd.hip), and the chunk plans of a batch of 16,384 (the shard
sizes of BASELINE configs[3] are 8,192 - 32,768 per GPU).  Tolerances as in test_gpu_parity.py (rtol 1e-3)."""
import pytest
import torch

import conftest
from conftest import assert_grads_close, assert_params_close, load_golden, sub, truth64
from test_cpu_host import _build

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def close(a, b, rtol=1e-3, atol=1e-4):
    torch.testing.assert_close(a.detach().float().cpu(), b.float(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("seq1d", [True, False], ids=["seq1d", "per_layer"])
def test_model_without_kinks_needs_no_hatch(seq1d):
    """TCVAE (C=3, N=3, L=60, first_kernel=21) with slope = 1.0 at B = 4096: LeakyReLU is the identity, so no
    pre-activation can flip sides and every deviation from the fp32 reference is rounding.  Run on the whole-model
    path (seq1d.h) and on the per-layer program; both must meet rtol 1e-3 element-wise with ZERO uses of the fp64
    hatch and of the kink hatch (which is closed here: kinks=False)."""
    import vae_amd
    from oracle import vae_oracle as O
    from vae_amd.Decoders import LongShort_TCVAE_Decoder
    from vae_amd.Encoders import LongShort_TCVAE_Encoder
    from vae_amd.train import FusedTrainStep
    B, kw = 4096, dict(num_layers=3, slope=1.0)
    torch.manual_seed(31)
    m = vae_amd.VariationalAutoencoder(input_size=3, hidden_size=30, num_layers=3, latent_dims=15,
                                       v_encoder=LongShort_TCVAE_Encoder, v_decoder=LongShort_TCVAE_Decoder,
                                       L=60, slope=1.0, first_kernel=21)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(32)
    x = 50 + 10 * torch.randn(B, 3, 60, generator=gen)
    eps = torch.randn(B, 30, generator=gen)
    ref = O.loss_and_grads("tcvae", sd, x, eps, **kw)
    m.to(DEV).train()
    st = FusedTrainStep(m, vae_amd.optim.Adam(m.parameters(), lr=1e-3), B, DEV, use_graph=False, allow_seq1d=seq1d)
    assert bool(st.prog.seq1d) == seq1d
    st.inject_eps = True
    st.eps.copy_(eps)
    loss = st.step(x.to(DEV))
    close(loss, ref["loss"].reshape(1), atol=1.0)
    close(st.output("x_rec"), ref["x_rec"])
    close(st.output("mu"), ref["mu"])
    close(st.output("logvar"), ref["logvar"])
    before = (conftest.HATCH["grad_fp64_hatch"], conftest.HATCH["grad_kink_hatch"])
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert_grads_close(grads, ref["grads"], truth=truth64("tcvae", sd, x, eps, **kw), kinks=False)
    after = (conftest.HATCH["grad_fp64_hatch"], conftest.HATCH["grad_kink_hatch"])
    assert after == before, f"a model without kinks used the tolerance hatches: {before} -> {after}"


_ORACLE = {}


def _mnist_case(B):
    """Oracle results for the notebook model on a seeded batch of B samples (cached: fp32 + fp64 passes on the CPU)."""
    from oracle import vae_oracle as O
    if B not in _ORACLE:
        g = load_golden("mnist_lat15")
        sd = sub(g, "sd/")
        gen = torch.Generator().manual_seed(41 + B)
        x = torch.rand(B, 1, 28, 28, generator=gen)
        eps = torch.randn(B, 15, generator=gen)
        ref = O.loss_and_grads("mnist", sd, x, eps)
        sd2 = {k: v.clone() for k, v in sd.items()}
        O.train_step("mnist", sd2, O.Adam(sd2, lr=1e-3), x, eps)
        _ORACLE[B] = (sd, x, eps, ref, truth64("mnist", sd, x, eps), sd2)
    return _ORACLE[B]


@pytest.mark.parametrize("B,fc", [(4096, "0"), (4096, "1"), (16384, None), (16384, "1"), (65536, None)],
                         ids=["b4096_fc_off", "b4096_fc_on", "b16384_auto", "b16384_fc_on", "b65536_auto"])
def test_mnist_parity_across_the_linear_path_switch(B, fc, monkeypatch):
    """The MNIST step against the oracle with the Linear/reparam run forced op by op and as fc_chain_k at the benched
    batch, and at B = 16,384, beyond FC_AUTO_MAX_BATCH, where the executor itself switches to the per-op Linear kernels
    (and the conv kernels run other chunk plans), and at B = 65,536, the batch of bench.py's strong_scaling_ref and the
    global batch of BASELINE configs[3]; the environment variable is read at every run of a program."""
    import vae_amd
    from vae_amd.train import FusedTrainStep
    if fc is None:
        monkeypatch.delenv("VAE_AMD_FC", raising=False)
    else:
        monkeypatch.setenv("VAE_AMD_FC", fc)
    sd, x, eps, ref, truth, sd_next = _mnist_case(B)
    m = _build("mnist_lat15")
    m.load_state_dict(sd)
    m.to(DEV).train()
    st = FusedTrainStep(m, vae_amd.optim.Adam(m.parameters(), lr=1e-3), B, DEV, use_graph=False)
    st.inject_eps = True
    st.eps.copy_(eps)
    loss = st.step(x.to(DEV))
    close(loss, ref["loss"].reshape(1), atol=1.0)
    out = st.prog.scalar_view(st.work, st.out_ref, 4).cpu()
    close(out[1], ref["recon"], atol=1.0)
    close(out[2], ref["kld"], atol=1.0)
    close(st.output("x_rec"), ref["x_rec"])
    close(st.output("mu"), ref["mu"])
    close(st.output("logvar"), ref["logvar"])
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    # (the last transposed convolution sits behind no ReLU: sigmoid follows it)
    assert_grads_close(grads, ref["grads"], truth=truth, kink_exempt=("decoder.decoder_conv.6",))
    new_sd = m.state_dict()
    for k, v in ref["new_stats"].items():
        close(new_sd[k], v, rtol=1e-4, atol=1e-5)
    assert_params_close({k: v.cpu() for k, v in new_sd.items()}, sd_next, ref["grads"], steps=1, lr=1e-3)
