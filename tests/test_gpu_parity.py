"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors the
reference produced and against the CPU oracle on fresh seeded inputs.

Tolerances (BASELINE.json north_star): forward outputs, ELBO terms and gradients within
rtol 1e-3 in fp32; gradients additionally get the cancellation floor documented in
conftest.assert_grads_close."""
import pytest
import torch

from conftest import assert_grads_close, assert_params_close, load_golden, sub, truth64
from test_cpu_host import _build

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
RTOL = 1e-3


def close(a, b, rtol=RTOL, atol=1e-4):
    torch.testing.assert_close(a.detach().float().cpu(), b.float(), rtol=rtol, atol=atol)


def _crit(family):
    import vae_amd
    return vae_amd.train.criterion_bce if family == "mnist" else vae_amd.train.criterion


def _target(family, x):
    return x if family == "mnist" else x[:, :, 0]


def test_forward_loss_grads_match_reference(golden_case):
    stem, family, kw, g = golden_case
    m = _build(stem)
    m.load_state_dict(sub(g, "sd/"))
    m.to(DEV).train()
    x = g["x"].to(DEV)
    x_rec, mu, logvar = m(x, eps=g["eps0"].to(DEV))
    close(x_rec, g["out/x_rec"])
    close(mu, g["out/mu"])
    close(logvar, g["out/logvar"])
    loss = _crit(family)(x_rec, _target(family, x), mu, logvar)
    close(loss, g["loss0"], atol=1e-2)
    loss.backward()
    got = {k: p.grad for k, p in m.named_parameters()}
    assert_grads_close(got, sub(g, "grad/"), truth=truth64(family, sub(g, "sd/"), g["x"], g["eps0"], **kw))
    sd = m.state_dict()
    for k, v in sub(g, "sd1/").items():
        close(sd[k], v, atol=1e-5)


def test_autograd_adam_trajectory(golden_case):
    """zero_grad / forward / criterion / backward / vae_amd.optim.Adam.step, 3 steps."""
    import vae_amd
    stem, family, kw, g = golden_case
    m = _build(stem)
    m.load_state_dict(sub(g, "sd/"))
    m.to(DEV).train()
    opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
    x = g["x"].to(DEV)
    for step in range(3):
        opt.zero_grad()
        x_rec, mu, logvar = m(x, eps=g[f"eps{step}"].to(DEV))
        loss = _crit(family)(x_rec, _target(family, x), mu, logvar)
        loss.backward()
        opt.step()
        close(loss, g[f"loss{step}"], rtol=2e-3, atol=1e-2)
    assert_params_close(m.state_dict(), sub(g, "sd3/"), sub(g, "grad/"), steps=3, lr=1e-3)


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_train_step_trajectory(golden_case, use_graph):
    """The fused {forward, ELBO, backward, Adam} step (eager and as a hipGraph replay)."""
    import vae_amd
    from vae_amd.train import FusedTrainStep
    stem, family, kw, g = golden_case
    m = _build(stem)
    m.load_state_dict(sub(g, "sd/"))
    m.to(DEV).train()
    opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
    x = g["x"].to(DEV)
    st = FusedTrainStep(m, opt, x.shape[0], DEV, use_graph=use_graph)
    st.inject_eps = True
    total = 0.0
    for step in range(3):
        st.eps.copy_(g[f"eps{step}"])
        loss = st.step(x)
        want = g[f"loss{step}"].item()
        assert abs(loss.item() - want) <= 2e-3 * abs(want) + 1e-2
        total += want
        if step == 0:
            close(st.output("x_rec"), g["out/x_rec"])
            got = {k: p.grad.clone() for k, p in m.named_parameters()}
            # p.grad aliases the fused gradient buffer of the LAST step: compare right after step 0
            assert_grads_close(got, sub(g, "grad/"), truth=truth64(family, sub(g, "sd/"), g["x"], g["eps0"], **kw))
    assert abs(st.total.item() - total) <= 2e-3 * abs(total)
    assert_params_close(m.state_dict(), sub(g, "sd3/"), sub(g, "grad/"), steps=3, lr=1e-3)


def test_eval_forward(golden_case):
    stem, family, kw, g = golden_case
    m = _build(stem)
    m.load_state_dict(sub(g, "sd3/"))
    m.to(DEV).eval()
    x = g["x"].to(DEV)
    with torch.no_grad():
        x_rec, mu, logvar = m(x, eps=g["eps_eval"].to(DEV))
        loss = _crit(family)(x_rec, _target(family, x), mu, logvar)
    close(x_rec, g["eval/x_rec"])
    close(mu, g["eval/mu"])
    close(logvar, g["eval/logvar"])
    close(loss, g["eval/loss"], atol=1e-2)


@pytest.mark.parametrize("stem", ["tcvae_c3_n3_l60_fk21", "tcvae_c3_n4_l60_fk15", "multiscale_c3_l60"])
def test_eval_pruned_decoder_equals_full_decoder(stem, monkeypatch):
    """Eval forward of the 1-D families (train.py:75-89): by default the decoder is evaluated at time index 0 only
    (Decoders.py:65 returns x[:, :, 0]) with every BatchNorm folded into the next layer's weights (seq1d.h, d.prune);
    VAE_SEQ1D_PRUNE=0 runs the full decoder phases.  Both must give the golden eval outputs, at the golden batch and at a
    ragged multi-chunk batch."""
    g = load_golden(stem)
    m = _build(stem)
    m.load_state_dict(sub(g, "sd3/"))
    m.to(DEV).eval()
    gen = torch.Generator().manual_seed(9)
    xb = 50 + 10 * torch.randn(777, *g["x"].shape[1:], generator=gen)
    eb = torch.randn(777, g["eps_eval"].shape[1], generator=gen)
    outs = {}
    for prune in ("1", "0"):
        monkeypatch.setenv("VAE_SEQ1D_PRUNE", prune)
        with torch.no_grad():
            x_rec, mu, logvar = m(g["x"].to(DEV), eps=g["eps_eval"].to(DEV))
            big = m(xb.to(DEV), eps=eb.to(DEV))
        close(x_rec, g["eval/x_rec"])
        close(mu, g["eval/mu"])
        close(logvar, g["eval/logvar"])
        outs[prune] = [t.clone() for t in big]
    for a, b in zip(outs["1"], outs["0"]):
        close(a, b.cpu(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("stem,family,kw,B", [
    ("mnist_lat15", "mnist", {}, 300),
    ("multiscale_c3_l60", "multiscale", {}, 517),
    ("tcvae_c3_n3_l60_fk21", "tcvae", dict(num_layers=3, slope=0.2), 1000),
])
def test_against_oracle_at_larger_batch(stem, family, kw, B):
    """Fresh seeded inputs at ragged batch sizes (exercise tile edges, multi-chunk reductions and
    split-K): HIP path vs the CPU oracle on identical weights, batch and eps."""
    from oracle import vae_oracle as O
    g = load_golden(stem)
    sd = sub(g, "sd/")
    gen = torch.Generator().manual_seed(7)
    if family == "mnist":
        x = torch.rand(B, 1, 28, 28, generator=gen)
    else:
        x = 50 + 10 * torch.randn(B, 3, 60, generator=gen)
    lat = g["eps0"].shape[1]
    eps = torch.randn(B, lat, generator=gen)
    ref = O.loss_and_grads(family, sd, x, eps, **kw)
    m = _build(stem)
    m.load_state_dict(sd)
    m.to(DEV).train()
    xd = x.to(DEV)
    x_rec, mu, logvar = m(xd, eps=eps.to(DEV))
    close(x_rec, ref["x_rec"])
    close(mu, ref["mu"])
    close(logvar, ref["logvar"])
    loss = _crit(family)(x_rec, _target(family, xd), mu, logvar)
    close(loss, ref["loss"], atol=1e-1)
    loss.backward()
    assert_grads_close({k: p.grad for k, p in m.named_parameters()}, ref["grads"], truth=truth64(family, sd, x, eps, **kw))


def test_bench_workload_at_full_size():
    """BASELINE configs[1] at its full size: the MNIST VAE, batch 4096, run exactly as bench.py runs it (the
    fused step replayed as a hipGraph), against the CPU oracle on the same weights, batch and eps: loss,
    every gradient, BatchNorm running statistics and the parameters after the Adam step."""
    import vae_amd
    from oracle import vae_oracle as O
    from vae_amd.train import FusedTrainStep
    B = 4096
    g = load_golden("mnist_lat15")
    sd = sub(g, "sd/")
    gen = torch.Generator().manual_seed(21)
    x = torch.rand(B, 1, 28, 28, generator=gen)
    eps = torch.randn(B, 15, generator=gen)
    ref = O.loss_and_grads("mnist", sd, x, eps)
    m = _build("mnist_lat15")
    m.load_state_dict(sd)
    m.to(DEV).train()
    opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
    st = FusedTrainStep(m, opt, B, DEV, use_graph=True)
    st.inject_eps = True
    st.eps.copy_(eps)
    loss = st.step(x.to(DEV))
    close(loss, ref["loss"].reshape(1), atol=1.0)
    close(st.output("x_rec"), ref["x_rec"])
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert_grads_close(grads, ref["grads"], truth=truth64("mnist", sd, x, eps))
    new_sd = m.state_dict()
    for k, v in ref["new_stats"].items():
        close(new_sd[k], v, rtol=1e-4, atol=1e-5)
    # one Adam step moves every parameter by at most lr; compare against the oracle's update
    sd2 = {k: v.clone() for k, v in sd.items()}
    opt_ref = O.Adam(sd2, lr=1e-3)
    O.train_step("mnist", sd2, opt_ref, x, eps)
    assert_params_close({k: v.cpu() for k, v in new_sd.items()}, sd2, ref["grads"], steps=1, lr=1e-3)


def test_widened_model_vs_oracle():
    """BASELINE configs[4]: the MNIST VAE with 4x conv channels (32/64/128) and latent 256.  Layers beyond the
    LDS-resident kernels' budgets run on the K-loop MFMA kernels of wide.h with a materialised BatchNorm; results
    must not depend on which path ran."""
    import vae_amd
    from oracle import vae_oracle as O
    B = 96
    torch.manual_seed(5)
    m = vae_amd.mnist_vae.VariationalAutoencoder(256, channels=(32, 64, 128), hidden=100)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(6)
    x = torch.rand(B, 1, 28, 28, generator=gen)
    eps = torch.randn(B, 256, generator=gen)
    ref = O.loss_and_grads("mnist", sd, x, eps)
    m.to(DEV).train()
    x_rec, mu, logvar = m(x.to(DEV), eps=eps.to(DEV))
    close(x_rec, ref["x_rec"])
    close(mu, ref["mu"])
    loss = _crit("mnist")(x_rec, x.to(DEV), mu, logvar)
    close(loss, ref["loss"], atol=1e-1)
    loss.backward()
    assert_grads_close({k: p.grad for k, p in m.named_parameters()}, ref["grads"], truth=truth64("mnist", sd, x, eps))


def test_batch_of_one():
    """Edge of the batch dimension: a single sample, in eval mode (running statistics) and in train mode
    (BatchNorm over the L > 1 positions of the one sample, as torch does), against golden / oracle."""
    from oracle import vae_oracle as O
    g = load_golden("tcvae_c3_n3_l60_fk21")
    m = _build("tcvae_c3_n3_l60_fk21")
    m.load_state_dict(sub(g, "sd3/"))
    m.to(DEV).eval()
    x1, e1 = g["x"][:1], g["eps_eval"][:1]
    with torch.no_grad():
        x_rec, mu, logvar = m(x1.to(DEV), eps=e1.to(DEV))
    close(x_rec, g["eval/x_rec"][:1])
    close(mu, g["eval/mu"][:1])
    m.train()
    want = O.tcvae_forward(sub(g, "sd3/"), x1, e1, num_layers=3, slope=0.2, training=True)
    x_rec, mu, logvar = m(x1.to(DEV), eps=e1.to(DEV))
    close(x_rec, want[0].detach())
    close(mu, want[1].detach())
    close(logvar, want[2].detach())


def test_submodule_calls_and_reparam():
    """utils.py in the reference calls model.encoder / model.decoder / reparametrization_trick directly."""
    g = load_golden("tcvae_c3_n3_l60_fk21")
    m = _build("tcvae_c3_n3_l60_fk21")
    m.load_state_dict(sub(g, "sd/"))
    m.to(DEV).train()
    x, eps = g["x"].to(DEV), g["eps0"].to(DEV)
    mu, logvar = m.encoder(x)
    close(mu, g["out/mu"])
    z = m.reparametrization_trick(mu, logvar, eps=eps)
    close(z, g["out/z"])
    close(m.decoder(z), g["out/x_rec"])
    z2 = m.reparametrization_trick(mu, logvar)  # device-drawn eps
    assert z2.shape == z.shape and torch.isfinite(z2).all()
    # gradient flows through the three separate calls like through forward()
    loss = m.decoder(z).sum() + 0.0 * mu.sum()
    loss.backward()
    assert all(p.grad is not None for p in m.decoder.parameters())


def test_device_randn_moments():
    from vae_amd.vae import device_randn
    rng = torch.tensor([1234, 0, 0], dtype=torch.int64, device=DEV)
    a = device_randn((1 << 20,), rng, DEV)
    b = device_randn((1 << 20,), rng, DEV)
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3
    assert abs((a * a * a).mean().item()) < 2e-2 and abs((a ** 4).mean().item() - 3) < 5e-2
    assert not torch.equal(a, b) and int(rng[1].item()) == 2 * (1 << 18)


def test_device_sliding_window_matches_reference_loader():
    """vae_window_gather / DeviceSlidingWindow vs DataLoader(slidingWindow(data, L)) (train.py:21-32): same
    batches in order, ragged last batch, shuffled epoch covers every window once, invalid start -> NaN."""
    from torch.utils.data import DataLoader
    from vae_amd.train import DeviceSlidingWindow, slidingWindow
    gen = torch.Generator().manual_seed(5)
    data = torch.randn(3, 500, generator=gen)
    Lw, bs = 60, 64
    ref = list(DataLoader(slidingWindow(data, Lw), batch_size=bs, shuffle=False))
    dev = DeviceSlidingWindow(data, Lw, batch_size=bs, device=DEV)
    assert len(dev) == len(ref) and len(dev.dataset) == 440
    got = list(dev)
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g.shape == r.shape and torch.equal(g.cpu(), r)     # a gather: bit-exact
    assert len(DeviceSlidingWindow(data, Lw, batch_size=bs, drop_last=True, device=DEV)) == 440 // bs
    sh = DeviceSlidingWindow(data, Lw, batch_size=bs, shuffle=True, device=DEV)
    firsts = torch.cat([b[:, 0, 0] for b in sh]).cpu()
    assert sorted(firsts.tolist()) == sorted(data[0, :440].tolist())
    edge = dev.gather(torch.tensor([440, 441, -1, 0]))
    assert torch.equal(edge[0].cpu(), data[:, 440:500]) and torch.isnan(edge[1]).all() and torch.isnan(edge[2]).all()
    assert torch.equal(edge[3].cpu(), data[:, :60])


def test_train_epoch_from_device_windows():
    """train() (train.py:51-72 signature) fed by the on-device window loader: runs the fused step per batch and
    returns the epoch's average loss; the first step's loss equals the oracle's on the same first batch."""
    import vae_amd
    from vae_amd.train import DeviceSlidingWindow, criterion, train
    torch.manual_seed(0)
    model = vae_amd.VariationalAutoencoder(input_size=3, hidden_size=30, num_layers=3, latent_dims=15,
                                           v_encoder=vae_amd.Encoders.LongShort_TCVAE_Encoder,
                                           v_decoder=vae_amd.Decoders.LongShort_TCVAE_Decoder,
                                           L=60, slope=0.2, first_kernel=21).to(DEV)
    opt = vae_amd.optim.Adam(model.parameters(), lr=1e-3)
    t = torch.arange(700, dtype=torch.float32)
    data = torch.stack([50 + 10 * torch.sin(t / 17 + c) + torch.randn(700) for c in range(3)])
    loader = DeviceSlidingWindow(data, 60, batch_size=128, drop_last=True, device=DEV)
    avg1 = train(model, loader, criterion, opt, DEV, 1)
    avg2 = train(model, loader, criterion, opt, DEV, 2)
    assert avg1 == avg1 and avg2 < avg1, (avg1, avg2)      # finite, and the ELBO goes down


@pytest.mark.parametrize("family", ["mnist", "multiscale"])
def test_backward_in_gradient_buckets_equals_one_pass(family):
    """vae_run_backward_range over the data-parallel step's two buckets (decoder + loss ops, then encoder ops)
    leaves the same GRAD buffer as one vae_run_backward."""
    import vae_amd
    from vae_amd.train import FusedTrainStep
    torch.manual_seed(3)
    if family == "mnist":
        model, x = vae_amd.mnist_vae.VariationalAutoencoder(15), torch.rand(200, 1, 28, 28)
    else:
        model, x = vae_amd.multiscale_vae.vae(3, 60, 15), 50 + 10 * torch.randn(200, 3, 60)
    model = model.to(DEV).train()
    # (allow_seq1d=False: the whole-model 1-D path runs its backward whole; the data-parallel step, which is what
    # uses the buckets, keeps the per-layer program)
    st = FusedTrainStep(model, vae_amd.optim.Adam(model.parameters(), lr=1e-3), 200, DEV, use_graph=False,
                        allow_seq1d=False)
    st.inject_eps = True
    st.x.copy_(x.to(DEV))
    st.eps.copy_(torch.randn(st.eps.shape, generator=torch.Generator().manual_seed(4)).to(DEV))
    ext = [st.x.view(200, -1), st.eps]
    buckets = st.prog.grad_buckets()
    assert buckets is not None
    st.prog.forward(st.work, st.flat, st.sflat, ext, True)
    st.prog.backward(st.work, st.flat, st.grads, st.sflat, ext)
    whole = st.grads.clone()
    st.grads.fill_(7.0)                                          # the first range must clear the buffer
    st.prog.forward(st.work, st.flat, st.sflat, ext, True)
    for i, (begin, end, lo, hi) in enumerate(buckets):
        st.prog.backward_range(st.work, st.flat, st.grads, st.sflat, ext, begin, end, i == 0)
        if i == 0:                                               # the decoder's bucket is complete after its range
            torch.testing.assert_close(st.grads[lo:hi], whole[lo:hi], rtol=1e-4, atol=1e-5 * float(whole.abs().max()))
    torch.testing.assert_close(st.grads, whole, rtol=1e-4, atol=1e-5 * float(whole.abs().max()))
    assert float(whole.abs().max()) > 0
