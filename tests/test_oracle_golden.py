"""Pins the CPU oracle (oracle/vae_oracle.py) to vectors the reference itself produced
(tools/make_golden.py).  CPU only."""
import torch

from conftest import assert_grads_close, assert_params_close, sub
from oracle import vae_oracle as O

RTOL, ATOL = 1e-4, 1e-5


def close(a, b, rtol=RTOL, atol=ATOL):
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


def test_forward_loss_grads(golden_case):
    stem, family, kw, g = golden_case
    sd = sub(g, "sd/")
    r = O.loss_and_grads(family, sd, g["x"], g["eps0"], **kw)
    for k in ("x_rec", "mu", "logvar", "z"):
        close(r[k], g["out/" + k])
    close(r["loss"], g["loss0"])
    close(r["kld"], g["out/kld"], rtol=1e-4, atol=1e-2)
    close(r["recon"], g["out/recon"], rtol=1e-4, atol=1e-2)
    assert_grads_close(r["grads"], sub(g, "grad/"))
    for k, v in sub(g, "sd1/").items():
        close(r["new_stats"][k], v)


def test_adam_trajectory_and_eval(golden_case):
    stem, family, kw, g = golden_case
    sd = {k: v.clone() for k, v in sub(g, "sd/").items()}
    opt = O.Adam(sd, lr=1e-3)
    for step in range(3):
        loss = O.train_step(family, sd, opt, g["x"], g[f"eps{step}"], **kw)
        assert abs(loss - g[f"loss{step}"].item()) <= 2e-4 * abs(g[f"loss{step}"].item())
    assert_params_close(sd, sub(g, "sd3/"), sub(g, "grad/"), steps=3, lr=1e-3)
    # eval-mode forward is pinned on the reference's own trained state (eval does not cancel the
    # +-lr drift of noise-gradient biases the way train-mode BatchNorm does)
    r = O.loss_and_grads(family, sub(g, "sd3/"), g["x"], g["eps_eval"], training=False, **kw)
    close(r["x_rec"], g["eval/x_rec"], rtol=1e-3, atol=1e-4)
    close(r["mu"], g["eval/mu"], rtol=1e-3, atol=1e-4)
    close(r["loss"], g["eval/loss"], rtol=1e-3, atol=1e-3)


def test_lin_size_matches_reference_shapes():
    # printed module trees in draft_pipe.ipynb:177-236: Linear(126,15), Linear(90,15), Linear(72,15), Linear(48,15)
    assert O.lin_size(60, 3) * 3 == 126
    assert O.lin_size(60, 3, 21) * 3 == 90
    assert O.lin_size(60, 4) * 3 == 72
    assert O.lin_size(60, 4, 15) * 3 == 48
    try:
        O.lin_size(8, 5)
    except ValueError:
        pass
    else:
        raise AssertionError("lin_size must raise when the window is too short")


def _saturated_bce_inputs():
    """x_rec with exact 0 / 1, denormals and near-1 values against targets 0 / 1 / 0.5: the log clamp at -100 and
    the max((1-p)p, 1e-12) of torch's backward (the reference calls F.binary_cross_entropy, notebook :143)."""
    ps = torch.tensor([0.0, 1.0, 1e-45, 1e-38, 1e-30, 1e-7, 0.5, 1 - 6e-8, 1 - 1e-3, 0.999999])
    ts = torch.tensor([0.0, 1.0, 0.5])
    p = ps.repeat_interleave(3).repeat(784 * 2 // 30 + 1)[:784 * 2].reshape(2, 784).contiguous()
    t = ts.repeat(10).repeat(784 * 2 // 30 + 1)[:784 * 2].reshape(2, 784).contiguous()
    return p, t


def test_bce_restatement_matches_torch_op_at_saturation():
    import torch.nn.functional as F
    p, t = _saturated_bce_inputs()
    mu, lv = torch.zeros(2, 15), torch.zeros(2, 15)
    p1 = p.clone().requires_grad_(True)
    want = F.binary_cross_entropy(p1, t, reduction="sum")
    (gw,) = torch.autograd.grad(want, p1)
    p2 = p.clone().requires_grad_(True)
    loss, recon, k = O.elbo_bce(p2, t, mu, lv)
    (gg,) = torch.autograd.grad(loss, p2)
    assert torch.isfinite(recon) and float(k) == 0.0
    close(recon, want, rtol=1e-6, atol=0)
    torch.testing.assert_close(gg, gw, rtol=1e-6, atol=0)
    assert float(gg.abs().max()) >= 1e11  # the saturated elements really are in the 1/1e-12 regime
