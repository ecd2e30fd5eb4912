"""fc.h: a run of Linear / reparameterisation ops as one launch (fc_chain_k) against the same ops launched one by one
(VAE_AMD_FC=0) and against torch autograd.  Both paths are fp32 MFMA with different summation orders: rtol 2e-5."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _build(B, K0, H, Lz, H2, N2, gen, two_branches=True, kl_grad=True):
    from vae_amd import _lib as L
    from vae_amd.program import ProgramBuilder

    def lin(n, k):
        return (torch.randn(n, k, generator=gen) / k ** 0.5).to(DEV), (0.1 * torch.randn(n, generator=gen)).to(DEV)
    shapes = [(H, K0), (H, K0), (Lz, H), (Lz, H), (H2, Lz), (N2, H2)]
    params = [lin(n, k) for n, k in shapes]
    offs, flat, o = {}, [], 0
    for w, b in params:
        for t in (w, b):
            offs[id(t)] = o
            o += (t.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned like the module's flat buffer
            flat.append(torch.cat([t.flatten(), torch.zeros((-t.numel()) % 4, device=DEV)]))
    flat = torch.cat(flat)
    b = ProgramBuilder(B, offs, {})
    x = b.input("x", K0, grad=True)
    eps = b.input("eps", Lz)
    mu, lv = b.act(Lz), b.act(Lz)
    (wa, ba), (wb, bb), (wm, bm), (wl, bl), (w1, b1), (w2, b2) = params
    ta = b.linear(x, wa, ba, K0, H, act=L.ACT_LEAKY, acc_din=two_branches)
    if two_branches:
        tb = b.linear(x, wb, bb, K0, H, act=L.ACT_LEAKY)
    else:
        tb = ta
    b.linear(ta, wm, bm, H, Lz, out=mu, acc_din=not two_branches)
    b.linear(tb, wl, bl, H, Lz, out=lv)
    z = b.reparam(mu, lv, eps, Lz, kl_grad=kl_grad)
    d1 = b.linear(z, w1, b1, Lz, H2, act=L.ACT_LEAKY)
    y = b.linear(d1, w2, b2, H2, N2)
    b.output("y", y, (N2,))
    return b.build(), flat, offs, params, (x, mu, lv, z, y)


def _run(prog, flat, bufs, xs, epss, dout, fc, ws_min_b=None):
    os.environ["VAE_AMD_FC"] = "1" if fc else "0"
    if ws_min_b is not None:
        os.environ["VAE_LINWS_MIN_B"] = str(ws_min_b)
    try:
        x, mu, lv, z, y = bufs
        work = prog.new_workspace(DEV)
        work.zero_()
        grads, state = torch.zeros_like(flat), torch.zeros(4, device=DEV)
        ext = [xs, epss]
        prog.forward(work, flat, state, ext, True)
        out = {k: prog.view(work, v, (v.n,)).clone() for k, v in (("mu", mu), ("lv", lv), ("z", z), ("y", y))}
        prog.grad_view(work, y).copy_(dout)
        prog.backward(work, flat, grads, state, ext)
        torch.cuda.synchronize()
        out["dx"] = prog.grad_view(work, x, (x.n,)).clone()
        out["grads"] = grads.clone()
        return out
    finally:
        os.environ.pop("VAE_AMD_FC", None)
        os.environ.pop("VAE_LINWS_MIN_B", None)


@pytest.mark.parametrize("B,K0,H,Lz,H2,N2,two", [
    (4096, 288, 100, 15, 100, 288, True),   # the notebook model's run
    (37, 51, 37, 7, 23, 45, True),          # nothing a multiple of 4, ragged last workgroup
    (16, 64, 32, 16, 32, 64, False),        # one branch feeding both heads (the second head accumulates d(t))
    (1, 20, 8, 4, 8, 12, True),
])
def test_chain_equals_op_by_op_and_autograd(B, K0, H, Lz, H2, N2, two):
    gen = torch.Generator().manual_seed(B * 7 + K0)
    prog, flat, offs, params, bufs = _build(B, K0, H, Lz, H2, N2, gen, two_branches=two)
    xs = torch.randn(B, K0, generator=gen).to(DEV)
    epss = torch.randn(B, Lz, generator=gen).to(DEV)
    dout = torch.randn(B, N2, generator=gen).to(DEV)
    a = _run(prog, flat, bufs, xs, epss, dout, fc=True)
    r = _run(prog, flat, bufs, xs, epss, dout, fc=False)
    for k in a:
        ref = r[k]
        torch.testing.assert_close(a[k], ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()), msg=lambda m, k=k: f"{k}: {m}")
    # torch autograd on the same numbers
    ps = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in params]
    (wa, ba), (wb, bb), (wm, bm), (wl, bl), (w1, b1), (w2, b2) = ps
    xt = xs.clone().requires_grad_(True)
    ta = F.relu(F.linear(xt, wa, ba))
    tb = F.relu(F.linear(xt, wb, bb)) if two else ta
    mu, lv = F.linear(ta, wm, bm), F.linear(tb, wl, bl)
    z = mu + epss * torch.exp(0.5 * lv)
    y = F.linear(F.relu(F.linear(z, w1, b1)), w2, b2)
    kld = -0.5 * torch.sum(1 + lv - mu ** 2 - lv.exp())  # the REPARAM op's kl_grad folds d(KLD) into d(mu), d(logvar)
    ((y * dout).sum() + kld).backward()

    def close(got, ref, what):
        torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-4 * float(ref.abs().max()), msg=lambda m: f"{what}: {m}")
    close(a["y"], y.detach(), "y")
    close(a["z"], z.detach(), "z")
    close(a["dx"], xt.grad, "dx")
    for i, ((w, b), (wt, bt)) in enumerate(zip(params, ps)):
        if not two and i == 1:
            continue  # the second branch does not exist in this program
        close(a["grads"][offs[id(w)]:offs[id(w)] + w.numel()].view_as(w), wt.grad, f"dW{i}")
        close(a["grads"][offs[id(b)]:offs[id(b)] + b.numel()], bt.grad, f"db{i}")


@pytest.mark.parametrize("B,K0,H,Lz,H2,N2,two", [
    (300, 288, 100, 16, 100, 288, True),    # the notebook model's shapes (latent 16: 16-byte rows); two column groups in d(x)
    (777, 64, 48, 8, 52, 100, False),       # ragged last row tile, column tiles that are not full
    (33, 32, 320, 4, 12, 36, True),         # 320 columns: two column groups forward; fewer rows than a CU has workgroups
])
def test_weights_stationary_linear_equals_the_tile_kernel(B, K0, H, Lz, H2, N2, two):
    """lin_ws_k (large batches: a column group of W stays in LDS, 16-row tiles stream) against lin_fwd_k on the same
    op-by-op program: forward values, input gradient, parameter gradients.  Different summation order: rtol 2e-5."""
    gen = torch.Generator().manual_seed(B * 11 + K0)
    prog, flat, offs, params, bufs = _build(B, K0, H, Lz, H2, N2, gen, two_branches=two)
    xs = torch.randn(B, K0, generator=gen).to(DEV)
    epss = torch.randn(B, Lz, generator=gen).to(DEV)
    dout = torch.randn(B, N2, generator=gen).to(DEV)
    a = _run(prog, flat, bufs, xs, epss, dout, fc=False, ws_min_b=1)
    r = _run(prog, flat, bufs, xs, epss, dout, fc=False, ws_min_b=1 << 30)
    for k in a:
        ref = r[k]
        torch.testing.assert_close(a[k], ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()), msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("B,K0,H,Lz,H2,N2,two", [
    (300, 288, 100, 16, 100, 288, True),    # the notebook model's shapes: 18 x 7 and 7 x 18 accumulator tiles
    (37, 51, 37, 7, 23, 45, True),          # nothing a multiple of 4 (scalar staging), a ragged only chunk
    (200, 64, 320, 4, 12, 36, False),       # 20 tiles in the split dimension, several chunks per slice is not reached: 4 slices
])
def test_whole_layer_weight_gradient_equals_the_tile_kernel(B, K0, H, Lz, H2, N2, two, monkeypatch):
    """lin_wgrad2_k (one workgroup owns a layer's whole dW for its batch slice; default from B = 8192) forced on at small
    batches against lin_wgrad_k: parameter gradients of every Linear layer.  Different summation order: rtol 2e-5."""
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("VAE_AMD_LINWG2", mode)  # (sizes the partial regions at build time and picks the kernel at run time)
        gen = torch.Generator().manual_seed(B * 13 + K0)
        prog, flat, offs, params, bufs = _build(B, K0, H, Lz, H2, N2, gen, two_branches=two)
        xs = torch.randn(B, K0, generator=gen).to(DEV)
        epss = torch.randn(B, Lz, generator=gen).to(DEV)
        dout = torch.randn(B, N2, generator=gen).to(DEV)
        res[mode] = _run(prog, flat, bufs, xs, epss, dout, fc=False)
    for k in res["1"]:
        ref = res["0"][k]
        torch.testing.assert_close(res["1"][k], ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()), msg=lambda m, k=k: f"{k}: {m}")
