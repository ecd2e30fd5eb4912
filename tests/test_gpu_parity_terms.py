"""GPU parity, second file: the ELBO TERMS (recon, kld) of the HIP path, the BCE saturation edge case, the
`test()` loop (train.py:75-89) and the 1-D model families at the benched batch size.  Same tolerances as
test_gpu_parity.py (BASELINE.json north_star: rtol 1e-3, fp32)."""
import ctypes

import pytest
import torch

from conftest import assert_grads_close, assert_params_close, load_golden, sub, truth64
from test_cpu_host import _build
from test_oracle_golden import _saturated_bce_inputs

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def close(a, b, rtol=1e-3, atol=1e-4):
    torch.testing.assert_close(a.detach().float().cpu(), b.float(), rtol=rtol, atol=atol)


def test_elbo_terms_of_fused_step_and_standalone_criterion(golden_case):
    """out[1] = recon and out[2] = kld (elem.h elbo_finalize_k) against the reference's own values
    (golden out/recon, out/kld: train.py:41-47 / notebook :143-150), from the fused train step and from
    vae_elbo_forward called on the reference's forward outputs."""
    import vae_amd
    from vae_amd import _lib as L
    from vae_amd.train import FusedTrainStep
    stem, family, kw, g = golden_case
    m = _build(stem)
    m.load_state_dict(sub(g, "sd/"))
    m.to(DEV).train()
    x = g["x"].to(DEV)
    st = FusedTrainStep(m, vae_amd.optim.Adam(m.parameters(), lr=1e-3), x.shape[0], DEV, use_graph=False)
    st.inject_eps = True
    st.eps.copy_(g["eps0"])
    st.step(x)
    out = st.prog.scalar_view(st.work, st.out_ref, 4).cpu()
    close(out[1], g["out/recon"], atol=1e-2)
    close(out[2], g["out/kld"], atol=1e-2)
    close(out[0], g["loss0"], atol=1e-2)
    # standalone criterion on the reference's own forward outputs
    xr = g["out/x_rec"].to(DEV).reshape(x.shape[0], -1).contiguous()
    tg = (g["x"] if family == "mnist" else g["x"][:, :, 0]).to(DEV).reshape(x.shape[0], -1).contiguous()
    mu, lv = g["out/mu"].to(DEV).contiguous(), g["out/logvar"].to(DEV).contiguous()
    res = torch.empty(4, device=DEV)
    part = torch.empty(2 * L.ELBO_BLOCKS, device=DEV)
    L.check(L.lib().vae_elbo_forward(L.ptr(xr), L.ptr(tg), L.ptr(mu), L.ptr(lv), L.ptr(res), L.ptr(part), x.shape[0],
                                     xr.shape[1], tg.stride(0), 1, mu.shape[1], int(family == "mnist"), L.stream_ptr()),
            "vae_elbo_forward")
    close(res[1], g["out/recon"], atol=1e-2)
    close(res[2], g["out/kld"], atol=1e-2)


def test_bce_saturated_probabilities():
    """x_rec holding exact 0.0 / 1.0 / denormals against targets 0 / 1 / 0.5: the log clamp at -100 of the
    forward (elem.h bce_term) and the max((1-p)p, 1e-12) of the backward (elbo_bwd_k) vs the oracle, which is
    pinned to torch's F.binary_cross_entropy on the same inputs (tests/test_oracle_golden.py)."""
    from oracle import vae_oracle as O
    from vae_amd.train import criterion_bce
    p, t = _saturated_bce_inputs()
    gen = torch.Generator().manual_seed(3)
    mu, lv = torch.randn(2, 15, generator=gen), 0.5 * torch.randn(2, 15, generator=gen)
    pr = p.clone().requires_grad_(True)
    mr, lr_ = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    want, wrecon, wk = O.elbo_bce(pr, t, mr, lr_)
    gp, gm, gl = torch.autograd.grad(want, (pr, mr, lr_))
    pd = p.to(DEV).requires_grad_(True)
    md, ld = mu.to(DEV).requires_grad_(True), lv.to(DEV).requires_grad_(True)
    loss = criterion_bce(pd.view(2, 1, 28, 28), t.to(DEV).view(2, 1, 28, 28), md, ld)
    assert torch.isfinite(loss)
    close(loss, want, rtol=1e-5, atol=1e-3)
    loss.backward()
    assert torch.isfinite(pd.grad).all()
    torch.testing.assert_close(pd.grad.cpu(), gp, rtol=1e-5, atol=1e-6)
    close(md.grad, gm, rtol=1e-5, atol=1e-6)
    close(ld.grad, gl, rtol=1e-5, atol=1e-6)


def test_test_loop_matches_reference_eval_loss(golden_case, monkeypatch, capsys):
    """vae_amd.train.test() (train.py:75-89): eval-mode BatchNorm, eps still drawn, criterion summed over the
    loader and divided by len(dataset).  eps is pinned by replacing the device draw with the golden eps_eval, the
    same way tools/make_golden.py pinned the reference's randn_like."""
    import vae_amd
    from torch.utils.data import DataLoader, TensorDataset
    stem, family, kw, g = golden_case
    m = _build(stem)
    m.load_state_dict(sub(g, "sd3/"))
    m.to(DEV)
    eps = g["eps_eval"].to(DEV)
    monkeypatch.setattr(vae_amd.vae, "device_randn", lambda shape, rng, device: eps.clone())
    loader = DataLoader(TensorDataset(g["x"]), batch_size=g["x"].shape[0])
    crit = vae_amd.train.criterion_bce if family == "mnist" else vae_amd.train.criterion
    got = vae_amd.train.test(m, loader, crit, DEV)
    want = float(g["eval/loss"]) / len(loader.dataset)
    assert abs(got - want) <= 1e-3 * abs(want) + 1e-3, (got, want)
    assert not m.training
    assert "====> Test set loss: {:.4f}".format(got) in capsys.readouterr().out


@pytest.mark.parametrize("stem,family,kw", [
    ("multiscale_c3_l60", "multiscale", {}),
    ("tcvae_c3_n3_l60_fk21", "tcvae", dict(num_layers=3, slope=0.2)),
])
def test_1d_models_at_benched_batch(stem, family, kw):
    """BASELINE configs[2] at its benched size: B = 4096 windows of (3, 60), run as bench.py runs it (fused step,
    hipGraph replay), against the CPU oracle: loss and its two terms, x_rec, every gradient, BatchNorm running
    statistics, parameters after the Adam step."""
    import vae_amd
    from oracle import vae_oracle as O
    from vae_amd.train import FusedTrainStep
    B = 4096
    g = load_golden(stem)
    sd = sub(g, "sd/")
    gen = torch.Generator().manual_seed(11)
    x = 50 + 10 * torch.randn(B, 3, 60, generator=gen)
    eps = torch.randn(B, 30, generator=gen)
    ref = O.loss_and_grads(family, sd, x, eps, **kw)
    m = _build(stem)
    m.load_state_dict(sd)
    m.to(DEV).train()
    opt = vae_amd.optim.Adam(m.parameters(), lr=1e-3)
    st = FusedTrainStep(m, opt, B, DEV, use_graph=True)
    st.inject_eps = True
    st.eps.copy_(eps)
    loss = st.step(x.to(DEV))
    close(loss, ref["loss"].reshape(1), atol=1.0)
    out = st.prog.scalar_view(st.work, st.out_ref, 4).cpu()
    close(out[1], ref["recon"], atol=1.0)
    close(out[2], ref["kld"], atol=1.0)
    close(st.output("x_rec"), ref["x_rec"])
    close(st.output("mu"), ref["mu"])
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert_grads_close(grads, ref["grads"], truth=truth64(family, sd, x, eps, **kw))
    new_sd = m.state_dict()
    for k, v in ref["new_stats"].items():
        close(new_sd[k], v, rtol=1e-4, atol=1e-5)
    sd2 = {k: v.clone() for k, v in sd.items()}
    O.train_step(family, sd2, O.Adam(sd2, lr=1e-3), x, eps, **kw)
    assert_params_close({k: v.cpu() for k, v in new_sd.items()}, sd2, ref["grads"], steps=1, lr=1e-3)
