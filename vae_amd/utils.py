"""Sampling helper of the reference's utils.py on the HIP path.

Only the model-evaluating part is mirrored (sample_from_data, utils.py:34-74); the matplotlib figures and sliders
around it (compare, experiment, ...) are out of scope and can keep calling this function."""
import torch


@torch.no_grad()
def sample_from_data(model, data, n, latent_dims, _eps=None):
    """n stochastic passes over every batch of `data` (a DataLoader over a slidingWindow dataset, unshuffled).

    Same arguments and return value as the reference (utils.py:35-74): numpy arrays x (T, C), the means over the n
    passes of mu, logvar, z (T, 2*latent_dims), x_rec reshaped to (T, n*C), and the tensor x_rec_mean (T, C).

    The reference runs the whole model n times per batch and draws a second eps for `z`
    (`model.reparametrization_trick`, utils.py:51-52).  Here, in eval mode, the n passes of a batch are ONE forward
    over the batch repeated n times along the batch dimension (BatchNorm uses running statistics in eval mode, so the
    samples do not interact) and one reparametrisation call over the stacked mu/logvar: 2 program runs per batch
    instead of 2n.  In training mode a BatchNorm couples the samples of a pass, so the passes run one by one as in
    the reference.  _eps = (eps_forward, eps_z), each (n, T, 2*latent_dims): injected noise (tests)."""
    latent = latent_dims * 2
    ds = data.dataset
    n_channels, T, B = ds.data.shape[0], ds.data.shape[1] - ds.L, data.batch_size
    dev = next(model.parameters()).device
    x = torch.empty(T, n_channels)
    mu, logvar, z = (torch.empty(n, T, latent) for _ in range(3))
    x_rec = torch.empty(n, T, n_channels)
    for i, batch in enumerate(data):
        lo = i * B
        b = batch.shape[0]
        xb = batch.to(dev, non_blocking=True)
        ef = ez = None
        if _eps is not None:
            ef, ez = (e[:, lo:lo + b].to(dev) for e in _eps)
        if not model.training:
            rep = xb.repeat(n, *([1] * (xb.dim() - 1)))
            rec, m, lv = model(rep, eps=None if ef is None else ef.reshape(n * b, latent))
            zz = model.reparametrization_trick(m, lv, eps=None if ez is None else ez.reshape(n * b, latent))
            for dst, src in ((mu, m), (logvar, lv), (z, zz), (x_rec, rec)):
                dst[:, lo:lo + b] = src.reshape(n, b, -1).cpu()
        else:
            for j in range(n):
                rec, m, lv = model(xb, eps=None if ef is None else ef[j])
                zz = model.reparametrization_trick(m, lv, eps=None if ez is None else ez[j])
                for dst, src in ((mu, m), (logvar, lv), (z, zz), (x_rec, rec)):
                    dst[j, lo:lo + b] = src.cpu()
        x[lo:lo + b] = batch[:, :, 0]
    mu, logvar, z, x_rec_mean = (torch.mean(t, dim=0) for t in (mu, logvar, z, x_rec))
    x_rec = torch.permute(x_rec, (1, 0, 2)).reshape(T, -1)
    x, mu, logvar, z, x_rec = (t.numpy() for t in (x, mu, logvar, z, x_rec))
    return x, mu, logvar, z, x_rec, x_rec_mean
