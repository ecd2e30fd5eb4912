"""Single-layer and conv -> BatchNorm(+ReLU) -> conv geometry sweep of the fused HIP kernels against torch's CPU
convolutions (fp32 reference of the same op): odd channel counts, kernel 1..5, stride 1..2, padding, output
padding, 1-D (H = 1) and 2-D, ragged batches.  Covers the instances the golden models do not reach (several
accumulator-tile buckets of the weight gradient, the vector-ALU variants, multi-class transposed gathers)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (cin, H, W, cout, kh, kw, stride, pad, transposed, out_pad, B)
GEOMS = [
    (1, 28, 28, 8, 3, 3, 2, 1, False, 0, 67),
    (3, 1, 60, 6, 1, 21, 2, 0, False, 0, 129),     # Conv1d-like, long first kernel
    (3, 1, 60, 6, 1, 2, 2, 0, False, 0, 33),
    (5, 9, 11, 7, 3, 2, 1, 1, False, 0, 40),        # odd everything, stride 1
    (8, 14, 14, 16, 3, 3, 2, 1, False, 0, 50),
    (16, 7, 7, 32, 3, 3, 2, 0, False, 0, 31),
    (24, 5, 5, 20, 5, 5, 1, 2, False, 0, 19),
    (4, 6, 6, 3, 1, 1, 1, 0, False, 0, 21),         # 1x1
    (32, 3, 3, 16, 3, 3, 2, 0, True, 0, 45),
    (16, 7, 7, 8, 3, 3, 2, 1, True, 1, 23),
    (8, 14, 14, 1, 3, 3, 2, 1, True, 1, 37),        # 1 output channel: vector-ALU variant
    (32, 14, 14, 1, 3, 3, 2, 1, True, 1, 29),       # 1 output channel, long K: still the vector-ALU variant
    (18, 1, 1, 12, 1, 2, 2, 0, True, 0, 64),        # TCVAE decoder ConvT1d k2 s2
    (6, 1, 4, 3, 1, 2, 2, 0, True, 0, 70),
    (7, 5, 4, 9, 4, 3, 2, 1, True, 0, 12),
    (3, 8, 8, 5, 3, 3, 1, 1, True, 0, 17),          # transposed, stride 1
    (2, 10, 10, 2, 5, 5, 2, 2, True, 1, 9),
    # GEMM + col2im kernel of scat.h (3x3 stride 2 transposed; the input gradient of a 3x3 stride-2 conv): odd channel
    # counts below the tile sizes, two samples per round, a ragged last round
    (12, 5, 6, 7, 3, 3, 2, 1, True, 1, 27),
    (20, 4, 4, 11, 3, 3, 2, 0, True, 0, 31),
    (6, 10, 10, 12, 3, 3, 2, 1, False, 0, 25),
    # wide layers: the K-loop MFMA kernels of wide.h (forward, input gradient and weight gradient)
    (32, 14, 14, 64, 3, 3, 2, 1, False, 0, 40),
    (64, 7, 7, 128, 3, 3, 2, 0, False, 0, 21),
    (128, 3, 3, 64, 3, 3, 2, 0, True, 0, 33),
    (64, 7, 7, 32, 3, 3, 2, 1, True, 1, 18),
    (20, 1, 30, 70, 1, 5, 1, 2, False, 0, 11),     # two channel tiles, ragged everything
    (40, 6, 5, 24, 3, 4, 2, 1, True, 1, 14),
    (48, 4, 4, 40, 2, 2, 2, 0, True, 0, 7),        # one tap per parity class
]


def _ref_conv(x, w, b, stride, pad, transposed, out_pad):
    if transposed:
        return F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=out_pad)
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def _tol(ref):
    return dict(rtol=1e-3, atol=2e-5 * max(float(ref.abs().max()), 1e-3))


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_single_conv_layer(geom):
    from vae_amd import _lib as L
    from vae_amd.program import ProgramBuilder
    cin, H, W, cout, kh, kw, stride, pad, transposed, out_pad, B = geom
    gen = torch.Generator().manual_seed(hash(geom) & 0xffff)
    w = torch.randn((cin, cout, kh, kw) if transposed else (cout, cin, kh, kw), generator=gen) * 0.3
    bias = torch.randn(cout, generator=gen)
    x = torch.randn(B, cin, H, W, generator=gen)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    y_ref = F.leaky_relu(_ref_conv(xr, wr, br, stride, (pad, pad) if H > 1 else (0, pad), transposed,
                                   (out_pad, out_pad) if H > 1 else (0, out_pad)), 0.2)
    dout = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(dout)

    wd, bd = w.to(DEV), bias.to(DEV)
    flat = torch.cat([wd.flatten(), bd])
    b = ProgramBuilder(B, {id(wd): 0, id(bd): wd.numel()}, {})
    xin = b.input("x", cin * H * W, grad=True)
    ph, oph = (pad, out_pad) if H > 1 else (0, 0)
    sh = stride if H > 1 else 1
    y, (OH, OW) = b.conv(xin, wd, bd, cin, (H, W), cout, (kh, kw), (sh, stride), (ph, pad), transposed=transposed,
                         out_pad=(oph, out_pad), act=L.ACT_LEAKY, slope=0.2)
    assert (OH, OW) == tuple(y_ref.shape[2:])
    b.output("y", y, (cout, OH, OW))
    prog = b.build()
    work = prog.new_workspace(DEV)
    work.zero_()
    grads, state = torch.zeros_like(flat), torch.zeros(4, device=DEV)
    ext = [x.to(DEV).view(B, -1)]
    prog.forward(work, flat, state, ext, True)
    got = prog.view(work, y, (cout, OH, OW)).cpu()
    torch.testing.assert_close(got, y_ref.detach(), **_tol(y_ref.detach()))
    prog.grad_view(work, y).copy_(dout.view(B, -1))
    prog.backward(work, flat, grads, state, ext)
    torch.cuda.synchronize()
    torch.testing.assert_close(prog.grad_view(work, xin, (cin, H, W)).cpu(), xr.grad, **_tol(xr.grad))
    torch.testing.assert_close(grads[:w.numel()].view(w.shape).cpu(), wr.grad, **_tol(wr.grad))
    torch.testing.assert_close(grads[w.numel():].cpu(), br.grad, **_tol(br.grad))


class _BN:
    def __init__(self, C, gen):
        self.weight = (1 + 0.2 * torch.randn(C, generator=gen)).to(DEV)
        self.bias = (0.2 * torch.randn(C, generator=gen)).to(DEV)
        self.running_mean = torch.zeros(C, device=DEV)
        self.running_var = torch.ones(C, device=DEV)


@pytest.mark.parametrize("geom", [
    # (cin, H, W, cmid, cout, k, stride, pad, transposed, out_pad, relu, B)
    (1, 28, 28, 8, 16, 3, 2, 1, False, 0, True, 48),
    (3, 1, 60, 6, 12, 2, 2, 0, False, 0, False, 130),
    (32, 3, 3, 16, 8, 3, 2, 1, True, 1, True, 26),
    (5, 6, 7, 9, 4, 3, 1, 1, False, 0, True, 35),
    (12, 1, 2, 6, 3, 2, 2, 0, True, 0, False, 77),
], ids=lambda g: "x".join(map(str, g)))
def test_conv_bn_conv_chain(geom):
    """The BatchNorm between two fused convs is virtual (statistics in the producer's epilogue, normalisation on
    the consumer's load, backward folded into both): forward, running statistics and every gradient vs torch."""
    from vae_amd import _lib as L
    from vae_amd.program import ProgramBuilder
    cin, H, W, cmid, cout, k, stride, pad, transposed, out_pad, relu, B = geom
    gen = torch.Generator().manual_seed(hash(geom) & 0xffff)
    kh = k if H > 1 else 1
    shp = (lambda a, c: (a, c, kh, k)) if transposed else (lambda a, c: (c, a, kh, k))
    w1, w2 = torch.randn(shp(cin, cmid), generator=gen) * 0.3, torch.randn(shp(cmid, cout), generator=gen) * 0.3
    b1, b2 = torch.randn(cmid, generator=gen), torch.randn(cout, generator=gen)
    bn = _BN(cmid, gen)
    x = 2.0 + torch.randn(B, cin, H, W, generator=gen)
    P = [t.clone().requires_grad_(True) for t in (x, w1, b1, bn.weight.cpu(), bn.bias.cpu(), w2, b2)]
    pd = (pad, pad) if H > 1 else (0, pad)
    op = (out_pad, out_pad) if H > 1 else (0, out_pad)
    st = (stride, stride) if H > 1 else (1, stride)
    conv = (lambda t, w, bb: F.conv_transpose2d(t, w, bb, stride=st, padding=pd, output_padding=op)) if transposed \
        else (lambda t, w, bb: F.conv2d(t, w, bb, stride=st, padding=pd))
    rm, rv = torch.zeros(cmid), torch.ones(cmid)
    h = F.batch_norm(conv(P[0], P[1], P[2]), rm, rv, P[3], P[4], training=True, momentum=0.1, eps=1e-5)
    h = F.relu(h) if relu else h
    y_ref = conv(h, P[5], P[6])
    dout = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(dout)

    dev = [t.to(DEV) for t in (w1, b1, w2, b2)]
    params = [dev[0], dev[1], bn.weight, bn.bias, dev[2], dev[3]]
    offs, o = {}, 0
    for t in params:
        offs[id(t)] = o
        o += t.numel()
    flat = torch.cat([t.flatten() for t in params])
    sflat = torch.cat([bn.running_mean, bn.running_var])
    b = ProgramBuilder(B, offs, {id(bn.running_mean): 0, id(bn.running_var): cmid})
    xin = b.input("x", cin * H * W, grad=True)
    hw, kk = (H, W), (kh, k)
    y1, hw1 = b.conv(xin, dev[0], dev[1], cin, hw, cmid, kk, st, pd, transposed=transposed, out_pad=op)
    n1 = b.batchnorm(y1, bn, cmid, hw1[0] * hw1[1], relu=relu)
    y2, hw2 = b.conv(n1, dev[2], dev[3], cmid, hw1, cout, kk, st, pd, transposed=transposed, out_pad=op)
    b.output("y", y2, (cout,) + tuple(hw2))
    prog = b.build()
    assert all(prog.ops[i].flags & L.F_FUSED for i in range(3)), "chain should take the fused path"
    work = prog.new_workspace(DEV)
    work.zero_()
    grads = torch.zeros_like(flat)
    ext = [x.to(DEV).view(B, -1)]
    prog.forward(work, flat, sflat, ext, True)
    torch.testing.assert_close(prog.view(work, y2, (cout,) + tuple(hw2)).cpu(), y_ref.detach(), **_tol(y_ref.detach()))
    torch.testing.assert_close(sflat[:cmid].cpu(), rm, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sflat[cmid:].cpu(), rv, rtol=1e-4, atol=1e-5)
    prog.grad_view(work, y2).copy_(dout.view(B, -1))
    prog.backward(work, flat, grads, sflat, ext)
    torch.cuda.synchronize()
    gmax = max(float(p.grad.abs().max()) for p in P[1:])
    torch.testing.assert_close(prog.grad_view(work, xin, (cin, H, W)).cpu(), P[0].grad, **_tol(P[0].grad))
    for t, ref in zip(params, P[1:]):
        got = grads[offs[id(t)]:offs[id(t)] + t.numel()].view(ref.shape).cpu()
        torch.testing.assert_close(got, ref.grad, rtol=1e-3, atol=2e-5 * gmax)
