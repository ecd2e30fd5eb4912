"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.  Not a product path.

A plain fp32 CPU restatement (torch.nn.functional on CPU tensors) of the reference's VAE
training hot path, written from the reference's behaviour, consuming a state_dict with
the reference's own keys.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; vae_amd/ never does (the product fails loudly when
its HIP library is missing).

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tools/make_golden.py produced by running the reference's own
modules in the build container (see that script's header).

Reference lines restated:
  lin_size                    Encoders.py:16-28, Decoders.py:16-28, multiscale_vae.py:26-29
  tcvae_encoder               Encoders.py:31-58 (ctor), :76-87 (forward)
  longshort encoder           Encoders.py:93-107
  tcvae_decoder               Decoders.py:35-54 (ctor), :57-65 (forward)
  reparam                     vae.py:15-18, multiscale_vae.py:206-209
  tcvae_forward               vae.py:20-23
  multiscale_forward          multiscale_vae.py:48-94, 96-141, 143-192, 211-220
  mnist_forward               .ipynb_checkpoints/vae-checkpoint.ipynb:31-69, 78-107, 115-130
  elbo_mse                    train.py:39-48
  elbo_bce                    .ipynb_checkpoints/vae-checkpoint.ipynb:141-151
  train step / Adam           train.py:55-67 + torch.optim.Adam defaults (draft_pipe.py:132)
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def lin_size(n, num_layers, first_kernel=None):
    """Encoders.py:16-28: window length after num_layers stride-2 convs, times 2*num_layers."""
    for i in range(num_layers):
        k = first_kernel if (i == 0 and first_kernel is not None) else 2
        n = 1 + (n - k) // 2
    if n <= 0:
        raise ValueError("Window Length is too small in relation to the number of Layers")
    return n * 2 * num_layers


def _bn(sd, prefix, x, training, new_stats):
    """BatchNorm1d/2d: train mode normalises with the biased batch variance and moves the
    running stats with momentum 0.1 using the UNBIASED variance; eval mode uses running stats."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    dims = [0] + list(range(2, x.dim()))
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        with torch.no_grad():
            new_stats[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            new_stats[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / max(n - 1, 1)
            new_stats[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = rm, rv
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * w.view(shape) + b.view(shape)


def _act(x, slope):
    return F.leaky_relu(x, slope)


def reparam(mu, logvar, eps):
    """vae.py:15-18 with eps injected instead of drawn."""
    return mu + eps * torch.exp(0.5 * logvar)


# ----------------------------------------------------------------------------- TCVAE (T)
def tcvae_encoder(sd, p, x, num_layers, slope, training, new_stats):
    """Encoders.py:76-87: N x [Conv1d s2 -> LeakyReLU -> BatchNorm1d], flatten, two Linear heads.
    Module indices in cnn_layers: conv 3i, act 3i+1, bn 3i+2 (Encoders.py:38-48)."""
    for i in range(num_layers):
        x = F.conv1d(x, sd[f"{p}.cnn_layers.{3 * i}.weight"], sd[f"{p}.cnn_layers.{3 * i}.bias"], stride=2)
        x = _act(x, slope)
        x = _bn(sd, f"{p}.cnn_layers.{3 * i + 2}", x, training, new_stats)
    h = x.reshape(x.shape[0], -1)
    mu = F.linear(h, sd[f"{p}.encoder_mu.0.weight"], sd[f"{p}.encoder_mu.0.bias"])
    logvar = F.linear(h, sd[f"{p}.encoder_logvar.0.weight"], sd[f"{p}.encoder_logvar.0.bias"])
    return mu, logvar


def tcvae_decoder(sd, p, z, num_layers, slope, training, new_stats):
    """Decoders.py:57-65: Linear -> (B, 2CN, 1) -> N x [ConvT1d k2 s2 -> LeakyReLU -> BN (not last)]
    -> time index 0 only.  cnn_layers indices: convT 3j, act 3j+1, bn 3j+2 (Decoders.py:45-54)."""
    x = F.linear(z, sd[f"{p}.decoder_lin.weight"], sd[f"{p}.decoder_lin.bias"]).unsqueeze(2)
    for j in range(num_layers):
        x = F.conv_transpose1d(x, sd[f"{p}.cnn_layers.{3 * j}.weight"], sd[f"{p}.cnn_layers.{3 * j}.bias"], stride=2)
        x = _act(x, slope)
        if j != num_layers - 1:
            x = _bn(sd, f"{p}.cnn_layers.{3 * j + 2}", x, training, new_stats)
    return x[:, :, 0]


def tcvae_forward(sd, x, eps, num_layers, slope=0.2, training=True):
    """vae.py:20-23 with LongShort encoder/decoder (Encoders.py:100-107, Decoders.py:77-78).
    Returns (x_rec, mu, logvar, z, new_stats)."""
    ns = {}
    smu, slv = tcvae_encoder(sd, "encoder.short_encoder", x, num_layers, slope, training, ns)
    lmu, llv = tcvae_encoder(sd, "encoder.long_encoder", x, num_layers, slope, training, ns)
    mu, logvar = torch.cat((smu, lmu), 1), torch.cat((slv, llv), 1)
    z = reparam(mu, logvar, eps)
    x_rec = tcvae_decoder(sd, "decoder.longshort_decoder", z, num_layers, slope, training, ns)
    return x_rec, mu, logvar, z, ns


# ------------------------------------------------------------------------ multiscale (S)
def _ms_encoder(sd, p, x, training, ns):
    for i in range(3):
        x = F.conv1d(x, sd[f"{p}.encoder_cnn.{3 * i}.weight"], sd[f"{p}.encoder_cnn.{3 * i}.bias"], stride=2)
        x = torch.relu(x)
        x = _bn(sd, f"{p}.encoder_cnn.{3 * i + 2}", x, training, ns)
    h = x.reshape(x.shape[0], -1)
    return (F.linear(h, sd[f"{p}.encoder_mu.0.weight"], sd[f"{p}.encoder_mu.0.bias"]),
            F.linear(h, sd[f"{p}.encoder_var_log.0.weight"], sd[f"{p}.encoder_var_log.0.bias"]))


def multiscale_forward(sd, x, eps, training=True):
    """multiscale_vae.py:211-220 (short k=2,2,2; long k=15,2,2; ReLU; channels C*2,4,8)."""
    ns = {}
    smu, slv = _ms_encoder(sd, "sencoder", x, training, ns)
    lmu, llv = _ms_encoder(sd, "lencoder", x, training, ns)
    mu, logvar = torch.cat((smu, lmu), 1), torch.cat((slv, llv), 1)
    z = reparam(mu, logvar, eps)
    h = F.linear(z, sd["decoder.dec_lin.0.weight"], sd["decoder.dec_lin.0.bias"]).unsqueeze(2)
    for j in range(3):
        h = F.conv_transpose1d(h, sd[f"decoder.dec_cnn.{3 * j}.weight"], sd[f"decoder.dec_cnn.{3 * j}.bias"], stride=2)
        h = torch.relu(h)
        if j != 2:
            h = _bn(sd, f"decoder.dec_cnn.{3 * j + 2}", h, training, ns)
    return h[:, :, 0], mu, logvar, z, ns


# ----------------------------------------------------------------------------- MNIST (M)
def mnist_forward(sd, x, eps, training=True):
    """Notebook Conv2d VAE (vae-checkpoint.ipynb:31-130).  Works for any channel widths the
    state_dict carries (the 'widened' BASELINE config uses the same keys with 4x channels)."""
    ns = {}
    e = "encoder.encoder_cnn"
    h = torch.relu(F.conv2d(x, sd[e + ".0.weight"], sd[e + ".0.bias"], stride=2, padding=1))
    h = F.conv2d(h, sd[e + ".2.weight"], sd[e + ".2.bias"], stride=2, padding=1)
    h = torch.relu(_bn(sd, e + ".3", h, training, ns))
    h = torch.relu(F.conv2d(h, sd[e + ".5.weight"], sd[e + ".5.bias"], stride=2, padding=0))
    h = h.reshape(h.shape[0], -1)

    def head(p):
        t = torch.relu(F.linear(h, sd[p + ".0.weight"], sd[p + ".0.bias"]))
        return F.linear(t, sd[p + ".2.weight"], sd[p + ".2.bias"])

    mu, logvar = head("encoder.encoder_mean"), head("encoder.encoder_std")
    z = reparam(mu, logvar, eps)
    d = torch.relu(F.linear(z, sd["decoder.decoder_lin.0.weight"], sd["decoder.decoder_lin.0.bias"]))
    d = F.linear(d, sd["decoder.decoder_lin.2.weight"], sd["decoder.decoder_lin.2.bias"])
    c = "decoder.decoder_conv"
    ch = sd[c + ".0.weight"].shape[0]
    d = d.reshape(d.shape[0], ch, 3, 3)
    d = F.conv_transpose2d(d, sd[c + ".0.weight"], sd[c + ".0.bias"], stride=2)
    d = torch.relu(_bn(sd, c + ".1", d, training, ns))
    d = F.conv_transpose2d(d, sd[c + ".3.weight"], sd[c + ".3.bias"], stride=2, padding=1, output_padding=1)
    d = torch.relu(_bn(sd, c + ".4", d, training, ns))
    d = F.conv_transpose2d(d, sd[c + ".6.weight"], sd[c + ".6.bias"], stride=2, padding=1, output_padding=1)
    return torch.sigmoid(d), mu, logvar, z, ns


# ------------------------------------------------------------------------------- losses
def kld(mu, logvar):
    return -0.5 * torch.sum(1 + logvar - mu * mu - torch.exp(logvar))


def elbo_mse(x_rec, target, mu, logvar):
    """train.py:39-48.  Returns (loss, recon, kld): sums, no mean, no beta."""
    recon = torch.sum((x_rec - target) ** 2)
    k = kld(mu, logvar)
    return recon + k, recon, k


class _BceSum(torch.autograd.Function):
    """F.binary_cross_entropy(p, t, reduction='sum') restated (notebook :143): the forward clamps each log at
    -100; the backward is torch's own formula (p - t) / max((1 - p) * p, 1e-12), NOT the derivative of the clamped
    logs (they differ where p saturates at 0 or 1).  tests/test_oracle_golden.py pins both against torch's op."""

    @staticmethod
    def forward(ctx, p, t):
        ctx.save_for_backward(p, t)
        logp = torch.clamp(torch.log(p), min=-100.0)
        log1mp = torch.clamp(torch.log(1 - p), min=-100.0)
        return -torch.sum(t * logp + (1 - t) * log1mp)

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        return g * (p - t) / torch.clamp((1 - p) * p, min=1e-12), None


def elbo_bce(x_rec, target, mu, logvar):
    """Notebook criterion (:141-151).  binary_cross_entropy clamps each log at -100."""
    p = x_rec.reshape(-1, 784)
    t = target.reshape(-1, 784)
    recon = _BceSum.apply(p, t)
    k = kld(mu, logvar)
    return recon + k, recon, k


# ---------------------------------------------------------------------------- train step
FAMILIES = {
    "tcvae": (tcvae_forward, elbo_mse, lambda x: x[:, :, 0]),
    "multiscale": (multiscale_forward, elbo_mse, lambda x: x[:, :, 0]),
    "mnist": (mnist_forward, elbo_bce, lambda x: x),
}


def is_param(key):
    return "running_" not in key and "num_batches" not in key


def loss_and_grads(family, sd, x, eps, training=True, **kw):
    """One forward + ELBO + backward.  Returns dict(out..., loss, recon, kld, grads, new_stats)."""
    fwd, crit, tgt = FAMILIES[family]
    leaves = {k: (v.detach().clone().requires_grad_(True) if is_param(k) else v) for k, v in sd.items()}
    x_rec, mu, logvar, z, ns = fwd(leaves, x, eps, training=training, **kw)
    loss, recon, k = crit(x_rec, tgt(x), mu, logvar)
    names = [n for n in leaves if is_param(n)]
    grads = torch.autograd.grad(loss, [leaves[n] for n in names]) if training else []
    return dict(x_rec=x_rec.detach(), mu=mu.detach(), logvar=logvar.detach(), z=z.detach(),
                loss=loss.detach(), recon=recon.detach(), kld=k.detach(),
                grads=dict(zip(names, grads)), new_stats=ns)


class Adam:
    """torch.optim.Adam defaults restated: betas (0.9, 0.999), eps 1e-8, no weight decay."""

    def __init__(self, sd, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in sd.items() if is_param(k)}
        self.v = {k: torch.zeros_like(v) for k, v in sd.items() if is_param(k)}

    def step(self, sd, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        for k, g in grads.items():
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            sd[k] = sd[k] - (self.lr / bc1) * self.m[k] / denom


def train_step(family, sd, opt, x, eps, **kw):
    """train.py:57-67 for one batch: forward, ELBO, backward, Adam, BN running-stat update.
    Mutates sd in place; returns the python float loss (the reference's loss.item())."""
    r = loss_and_grads(family, sd, x, eps, training=True, **kw)
    opt.step(sd, r["grads"])
    sd.update(r["new_stats"])
    return float(r["loss"])
