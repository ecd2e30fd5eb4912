"""MI355X drop-ins for train.py: slidingWindow, criterion, train, test (train.py:21-89), the
notebook's BCE criterion (vae-checkpoint.ipynb:141-151) and the fused train step they use.

Signatures are the reference's.  `train()` keeps its loop shape (per batch: to(device),
zero_grad, forward, criterion(x_rec, data[:,:,0], mu, logvar), backward, optimizer.step) but,
when it is handed a vae_amd model + vae_amd criterion + vae_amd.optim.Adam, runs each batch as
ONE hipGraph replay of {eps draw, forward, ELBO, backward, Adam} (FusedTrainStep) and reads the
loss back only at the reference's print cadence instead of every step (train.py:67).
"""
import ctypes

import torch
from torch.utils.data import Dataset

from . import _lib as L
from .optim import Adam
from .program import ProgramBuilder
from .vae import HipVAE


class slidingWindow(Dataset):
    """item i -> data[:, i:i+L]; len = T - L (train.py:21-32)."""

    def __init__(self, data, L):
        self.data = data
        self.L = L

    def __getitem__(self, index):
        if self.data.shape[1] - index >= self.L:
            return self.data[:, index:index + self.L]

    def __len__(self):
        return self.data.shape[1] - self.L


class DeviceSlidingWindow:
    """`DataLoader(slidingWindow(data, L), batch_size, shuffle)` with the series resident on the GPU.

    Yields (B, C, L) device batches gathered by one HIP kernel (vae_window_gather) from the (C, T) series, so an
    epoch moves T*C floats over PCIe once instead of (T-L)*C*L per epoch, and no collate runs on the host.
    Iteration order: `arange(T-L)`, or `torch.randperm` on the device when shuffle (same distribution as the
    DataLoader's RandomSampler, not the same permutation).  `.dataset` is the reference Dataset (train.py:21-32),
    so `train()` / `test()` print and normalise by `len(loader.dataset)` as before.
    """

    def __init__(self, data, L, batch_size=1, shuffle=False, drop_last=False, device="cuda", generator=None):
        data = torch.as_tensor(data, dtype=torch.float32)
        if data.dim() != 2 or data.shape[1] <= L:
            raise ValueError(f"DeviceSlidingWindow: need a (C, T) series with T > L, got {tuple(data.shape)}, L={L}")
        self.dataset = slidingWindow(data, L)
        self.series = data.to(device).contiguous()
        self.L, self.batch_size, self.shuffle, self.drop_last = L, int(batch_size), shuffle, drop_last
        self.generator = generator
        self.n = data.shape[1] - L

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def gather(self, starts, out=None):
        """starts: int64 window indices (device) -> (len(starts), C, L) batch; `out` may be a preallocated buffer."""
        L.require_device(self.series)
        starts = starts.to(device=self.series.device, dtype=torch.int64).contiguous()
        C, T = self.series.shape
        B = starts.numel()
        if out is None:
            out = torch.empty(B, C, self.L, dtype=torch.float32, device=self.series.device)
        assert out.is_contiguous() and out.numel() == B * C * self.L and out.dtype == torch.float32
        L.check(L.lib().vae_window_gather(self.series.data_ptr(), C, T, self.L, starts.data_ptr(), B, out.data_ptr(),
                                          L.stream_ptr()), "vae_window_gather")
        return out

    def __iter__(self):
        dev = self.series.device
        order = torch.randperm(self.n, device=dev, generator=self.generator) if self.shuffle \
            else torch.arange(self.n, device=dev)
        for i in range(len(self)):
            yield self.gather(order[i * self.batch_size:(i + 1) * self.batch_size])


# ------------------------------------------------------------------------------- criterion
class _ElboFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recon_x, x, mu, logvar, bce):
        L.require_device(recon_x)
        B = recon_x.shape[0]
        xr = recon_x.detach().float().contiguous().view(B, -1)
        tg = x.detach().float()
        tg = tg if tg.dim() == 2 else tg.reshape(B, -1)
        if tg.shape != xr.shape:
            raise ValueError(f"criterion: reconstruction {tuple(recon_x.shape)} vs target {tuple(x.shape)}")
        if tg.stride(0) < 0 or tg.stride(1) < 0:
            tg = tg.contiguous()
        m, lv = mu.detach().float().contiguous(), logvar.detach().float().contiguous()
        out = torch.empty(4, device=xr.device)
        partials = torch.empty(2 * L.ELBO_BLOCKS, device=xr.device)
        L.check(L.lib().vae_elbo_forward(L.ptr(xr), L.ptr(tg), L.ptr(m), L.ptr(lv), L.ptr(out), L.ptr(partials), B,
                                         xr.shape[1], tg.stride(0), tg.stride(1), m.shape[1], int(bce), L.stream_ptr()),
                "vae_elbo_forward")
        ctx.saved = (xr, tg, m, lv, bce, recon_x.shape)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        xr, tg, m, lv, bce, shape = ctx.saved
        B = xr.shape[0]
        gs = g.detach().float().reshape(1).contiguous()
        dxr, dmu, dlv = torch.empty_like(xr), torch.empty_like(m), torch.empty_like(lv)
        L.check(L.lib().vae_elbo_backward(L.ptr(xr), L.ptr(tg), L.ptr(m), L.ptr(lv), L.ptr(gs), L.ptr(dxr), L.ptr(dmu),
                                          L.ptr(dlv), B, xr.shape[1], tg.stride(0), tg.stride(1), m.shape[1], int(bce),
                                          L.stream_ptr()), "vae_elbo_backward")
        return dxr.view(shape), None, dmu, dlv, None


def criterion(recon_x, x, mu, logvar):
    """MSE-sum reconstruction + KL-sum (train.py:39-48); returns a 0-dim tensor."""
    return _ElboFn.apply(recon_x, x, mu, logvar, False)


def criterion_bce(recon_x, x, mu, logvar):
    """BCE-sum reconstruction + KL-sum, the notebook's criterion (vae-checkpoint.ipynb:141-151)."""
    return _ElboFn.apply(recon_x, x, mu, logvar, True)


criterion.elbo_bce = False
criterion_bce.elbo_bce = True


# ------------------------------------------------------------------------- fused train step
class FusedTrainStep:
    """{eps ~ N(0,1), forward, ELBO, backward, Adam} for one batch as a single captured hipGraph.

    Inputs are copied into static device buffers (graph nodes hold fixed addresses); loss of
    the last step and the running sum live in device memory (`loss`, `total`)."""

    def __init__(self, model, optimizer, B, device, bce=None, use_graph=True, allow_seq1d=True):
        if not isinstance(model, HipVAE) or not isinstance(optimizer, Adam):
            raise TypeError("FusedTrainStep needs a vae_amd VAE model and vae_amd.optim.Adam")
        self.model, self.opt, self.B, self.device = model, optimizer, B, torch.device(device)
        self.bce = model.elbo_bce if bce is None else bool(bce)
        self.group = optimizer.param_groups[0]
        self.dtype = L.get_compute_dtype()  # the program (and a captured graph) keeps the kernels of this dtype
        flat, sflat = model.pack(self.device)
        plist = [p for _, p in model.named_parameters()]
        if len(optimizer.param_groups) != 1 or [id(p) for p in self.group["params"]] != [id(p) for p in plist]:
            raise ValueError("fused step: the optimizer must hold exactly model.parameters() in one group")
        self.flat, self.sflat = flat, sflat
        self.m, self.v, self.tick = optimizer.flat_state(self.group, flat)
        self.grads = torch.zeros_like(flat)
        self._plist = plist
        self.bind_grads()
        b = ProgramBuilder(B, model._param_off, model._state_off, want_grads=True, nflat=flat.numel(),
                           allow_seq1d=allow_seq1d)
        x = b.input("x", model._numel(model.in_shape))
        eps = b.input("eps", model.latent_total)
        mu, logvar = b.act(model.latent_total), b.act(model.latent_total)
        model.emit_encoder(b, x, mu, logvar)
        z = b.reparam(mu, logvar, eps, model.latent_total, kl_grad=True)
        x_rec = model.emit_decoder(b, z)
        t_sb, t_se = model.target_strides()
        self.out_ref = b.elbo(x_rec, x.ref, t_sb, t_se, mu, logvar, model.latent_total, self.bce, loss_in_bwd=True)
        self.prog = b.build()
        self.bufs = dict(x_rec=(x_rec, model.out_shape), mu=(mu, (model.latent_total,)),
                         logvar=(logvar, (model.latent_total,)))
        self.work = self.prog.new_workspace(self.device)
        self.work.zero_()
        self.prog._clean_ptr = self.work.data_ptr()  # BatchNorm accumulators: zero
        self.x = torch.zeros((B,) + tuple(model.in_shape), device=self.device)
        self.eps = torch.zeros(B, model.latent_total, device=self.device)
        out = self.prog.scalar_view(self.work, self.out_ref, 4)
        self.loss, self.total = out[0:1], out[3:4]
        self.rng = model._rng_state(self.device)
        # Adam's {lr, beta1, beta2, eps} live in device memory and are re-read by the kernel at every replay, so a
        # captured graph follows optimizer.param_groups[0] (LR schedulers) without a re-capture
        self.hyper = torch.zeros(4, device=self.device)
        self._hyper_host = None
        self.inject_eps = False  # tests set this and fill self.eps themselves
        self._eps_ready = False  # self.eps holds the noise drawn for the coming step
        self.graph = None
        self.use_graph = use_graph
        self.steps = 0

    def bind_grads(self):
        """p.grad = view of this stepper's flat gradient buffer (stays meaningful for callers)."""
        for p in self._plist:
            o = self.model._param_off[id(p)]
            p.grad = self.grads[o:o + p.numel()].view(p.shape)

    def _enqueue(self, stream):
        # the noise of a step is drawn by the Adam launch of the step before it (no launch of its own); the first step
        # after construction, or after the caller injected its own eps, draws it here, outside any captured graph
        ext = [self.x.view(self.B, -1), self.eps]
        # (step() has made sure the BatchNorm accumulators of the workspace are zero: the forward pass then runs without
        # a memset node, and the backward pass, which begins with the ELBO op, clears them again for the next step)
        self.prog.forward(self.work, self.flat, self.sflat, ext, True, stream=stream)
        draw = not self.inject_eps
        self.prog.backward_step(self.work, self.flat, self.grads, self.sflat, ext, self.m, self.v, self.hyper, self.tick,
                                stream=stream, next_eps=self.eps if draw else None, rng=self.rng if draw else None)

    def _first_eps(self):
        if self.inject_eps:
            self._eps_ready = False
        elif not self._eps_ready:
            L.check(L.lib().vae_randn(L.ptr(self.eps), self.eps.numel(), L.ptr(self.rng), L.stream_ptr()), "vae_randn")
            self._eps_ready = True

    def _refresh_hyper(self):
        g = self.group
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
        if cur != self._hyper_host:
            self.hyper.copy_(torch.tensor(cur, dtype=torch.float32))  # ordered on the current stream, before the step
            self._hyper_host = cur

    def _capture(self):
        cs = ctypes.c_void_p()
        L.check(L.lib().vae_graph_begin(ctypes.byref(cs)), "vae_graph_begin")
        try:
            self._enqueue(cs)
        finally:
            h = ctypes.c_void_p()
            rc = L.lib().vae_graph_end(ctypes.byref(h))
        L.check(rc, "vae_graph_end")
        self.graph = h

    def step(self, x=None):
        """Runs one train step on batch x (None: reuse the resident batch).  Returns the device
        1-element loss tensor of this step (overwritten by the next step)."""
        if x is not None:
            if x.shape != self.x.shape:
                raise ValueError(f"fused step was built for batches {tuple(self.x.shape)}, got {tuple(x.shape)}")
            self.x.copy_(x, non_blocking=True)
        if L.get_compute_dtype() != self.dtype:
            raise L.VaeAmdError(f"this fused step was built under compute dtype {self.dtype}; "
                                f"the current one is {L.get_compute_dtype()} (build a new stepper)")
        self._refresh_hyper()
        self._first_eps()
        self.prog.ensure_clean(self.work)  # (a no-op unless someone ran other passes over this workspace)
        if self.use_graph:
            if self.graph is None or self._captured_inject != self.inject_eps:
                if self.graph is not None:
                    L.lib().vae_graph_destroy(self.graph)
                self._captured_inject = self.inject_eps
                torch.cuda.current_stream().synchronize()
                self._capture()
            L.check(L.lib().vae_graph_launch(self.graph, L.stream_ptr()), "vae_graph_launch")
        else:
            self._enqueue(L.stream_ptr())
        self.steps += 1
        for m in self.model._bn_holders():
            m._pending += 1
        return self.loss

    def output(self, name):
        buf, shape = self.bufs[name]
        return self.prog.view(self.work, buf, shape)

    def __del__(self):
        try:
            if self.graph is not None:
                L.lib().vae_graph_destroy(self.graph)
        except Exception:
            pass


def _fused_stepper(model, criterion_fn, optimizer, data):
    """Returns a FusedTrainStep when model/criterion/optimizer are all vae_amd's, else None."""
    if not (isinstance(model, HipVAE) and isinstance(optimizer, Adam) and criterion_fn in (criterion, criterion_bce)):
        return None
    # one stepper per (batch size, optimizer, criterion, device): a loader without drop_last alternates between the
    # full and the ragged last batch, and each keeps its program, workspace and hipGraph.  The flat parameter
    # buffer's address is part of the key, so a repack (model.cpu().cuda(), load of unpacked weights) starts afresh.
    flat, _ = model.pack(data.device)
    key = (data.shape[0], id(optimizer), criterion_fn.elbo_bce, str(data.device), flat.data_ptr(), L.get_compute_dtype())
    cache = model.__dict__.setdefault("_stepper", {})
    st = cache.get(key)
    if st is None:
        for k in [k for k in cache if k[4] != key[4]]:  # steppers of a stale flat buffer train dead storage
            del cache[k]
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        st = cache[key] = FusedTrainStep(model, optimizer, data.shape[0], data.device, bce=criterion_fn.elbo_bce)
    else:  # p.grad views follow the stepper that ran last
        st.bind_grads()
    return st


def _unpack(batch):
    return batch[0] if isinstance(batch, (tuple, list)) else batch  # notebook loaders yield (data, label)


def train(model, train_loader, criterion, optimizer, device, epoch):
    """One epoch (train.py:51-72).  Prints the reference's progress lines."""
    model.train()
    total_dev, total_host = None, 0.0
    n_batches = len(train_loader)
    for batch_idx, batch in enumerate(train_loader):
        data = _unpack(batch).to(device, non_blocking=True)
        stepper = _fused_stepper(model, criterion, optimizer, data)
        if stepper is not None:
            loss = stepper.step(data)
            total_dev = loss.clone() if total_dev is None else total_dev.add_(loss)
        else:  # generic autograd path (foreign model, criterion or optimizer)
            optimizer.zero_grad()
            x_rec, mu, logvar = model(data)
            target = data[:, :, 0] if data.dim() == 3 else data
            loss = criterion(x_rec, target, mu, logvar)
            loss.backward()
            optimizer.step()
            total_dev = loss.detach().clone().reshape(1) if total_dev is None else total_dev.add_(loss.detach())
        if batch_idx % 100 == 0:
            print('Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {:.6f}'.format(
                epoch, batch_idx * len(data), len(train_loader.dataset),
                100. * batch_idx / n_batches, loss.item() / len(data)))
    if total_dev is not None:
        total_host = float(total_dev.item())
    print('====> Epoch: {} Average loss: {:.4f}'.format(epoch, total_host / len(train_loader.dataset)))
    return total_host / len(train_loader.dataset)


def test(model, test_loader, criterion, device):
    """Eval-mode loss per sample (train.py:75-89): BatchNorm uses running stats, eps still drawn."""
    model.eval()
    total = None
    with torch.no_grad():
        for batch in test_loader:
            data = _unpack(batch).to(device, non_blocking=True)
            x_rec, mu, logvar = model(data)
            target = data[:, :, 0] if data.dim() == 3 else data
            loss = criterion(x_rec, target, mu, logvar)
            total = loss.detach().clone() if total is None else total.add_(loss)
    test_loss = (float(total.item()) if total is not None else 0.0) / len(test_loader.dataset)
    print('====> Test set loss: {:.4f}'.format(test_loss))
    return test_loss
