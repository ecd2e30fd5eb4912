"""MI355X drop-in for multiscale_vae.py: the hard-coded 3-layer short/long Conv1d VAE
(multiscale_vae.py:26-220).  Differences from the generic TCVAE, all kept: channels C*2, C*4,
C*8; ReLU; long first kernel 15; heads named encoder_var_log; lin_size(n, kernels)."""
import torch.nn as nn

from . import _lib as L
from .module import BatchNormParams, ConvParams, HipModule, LinearParams, NoParams
from .vae import HipVAE


def lin_size(n, kernels):
    """multiscale_vae.py:26-29: length after one stride-2 conv per kernel size."""
    for k in kernels:
        n = 1 + (n - k) // 2
    return n


class _MsEncoder(HipModule):
    kernels = (2, 2, 2)

    def __init__(self, n_channels, L, latent_dims):
        super().__init__()
        C = n_channels
        self.n_channels, self.L, self.latent_dims = C, L, latent_dims
        chans = (C, 2 * C, 4 * C, 8 * C)
        layers = []
        for i, k in enumerate(self.kernels):
            layers += [ConvParams((chans[i + 1], chans[i], k), chans[i + 1], chans[i] * k), NoParams(),
                       BatchNormParams(chans[i + 1])]
        self.encoder_cnn = nn.ModuleList(layers)
        self.n = lin_size(L, self.kernels)
        if self.n <= 0:
            raise ValueError("Window Length is too small in relation to the number of Layers")
        self.encoder_mu = nn.ModuleList([LinearParams(self.n * C * 8, latent_dims)])
        self.encoder_var_log = nn.ModuleList([LinearParams(self.n * C * 8, latent_dims)])

    def emit(self, b, x, mu_out, logvar_out):
        h, length = x, self.L
        for i in range(3):
            conv, bn = self.encoder_cnn[3 * i], self.encoder_cnn[3 * i + 2]
            cout, cin, k = conv.weight.shape
            h, (_, length) = b.conv(h, conv.weight, conv.bias, cin, (1, length), cout, (1, k), (1, 2), (0, 0),
                                    act=L.ACT_LEAKY, slope=0.0)
            h = b.batchnorm(h, bn, cout, length)
        K = self.n * self.n_channels * 8
        b.linear(h, self.encoder_mu[0].weight, self.encoder_mu[0].bias, K, self.latent_dims, out=mu_out, acc_din=True)
        b.linear(h, self.encoder_var_log[0].weight, self.encoder_var_log[0].bias, K, self.latent_dims, out=logvar_out)

    def emit_program(self, b, training):
        x = b.input("x", self.n_channels * self.L)
        mu, logvar = b.act(self.latent_dims), b.act(self.latent_dims)
        self.emit(b, x, mu, logvar)
        b.output("mu", mu, (self.latent_dims,))
        b.output("var_log", logvar, (self.latent_dims,))

    def forward(self, x):
        out = self.run_program([x])
        return out["mu"], out["var_log"]


class shortEncoder(_MsEncoder):
    """multiscale_vae.py:48-94 (kernels 2, 2, 2)."""
    kernels = (2, 2, 2)


class longEncoder(_MsEncoder):
    """multiscale_vae.py:96-141 (kernels 15, 2, 2)."""
    kernels = (15, 2, 2)


class Decoder(HipModule):
    """multiscale_vae.py:143-192: Linear(2*latent -> 8C), 3 x ConvTranspose1d k2 s2 + ReLU
    (BatchNorm after the first two), output = time index 0."""

    def __init__(self, n_channels, L, latent_dims):
        super().__init__()
        C = n_channels
        self.n_channels, self.latent_dims = C, latent_dims
        self.n = lin_size(L, [2, 2, 2])
        self.dec_lin = nn.ModuleList([LinearParams(2 * latent_dims, C * 8)])
        self.dec_cnn = nn.ModuleList([
            ConvParams((8 * C, 4 * C, 2), 4 * C, 4 * C * 2), NoParams(), BatchNormParams(4 * C),
            ConvParams((4 * C, 2 * C, 2), 2 * C, 2 * C * 2), NoParams(), BatchNormParams(2 * C),
            ConvParams((2 * C, C, 2), C, C * 2), NoParams()])

    def emit(self, b, z):
        C = self.n_channels
        h = b.linear(z, self.dec_lin[0].weight, self.dec_lin[0].bias, 2 * self.latent_dims, 8 * C)
        length = 1
        for j in range(3):
            conv = self.dec_cnn[3 * j]
            cin, cout, _ = conv.weight.shape
            h, (_, length) = b.conv(h, conv.weight, conv.bias, cin, (1, length), cout, (1, 2), (1, 2), (0, 0),
                                    transposed=True, act=L.ACT_LEAKY, slope=0.0)
            if j != 2:
                h = b.batchnorm(h, self.dec_cnn[3 * j + 2], cout, length)
        return b.slice0(h, C, length)

    def emit_program(self, b, training):
        z = b.input("z", 2 * self.latent_dims, grad=True)
        b.output("x_rec", self.emit(b, z), (self.n_channels,))

    def forward(self, x):
        return self.run_program([x])["x_rec"]


class vae(HipVAE):
    """multiscale_vae.py:198-220."""

    def __init__(self, n_channels, L, latent_dims):
        super().__init__()
        self.sencoder = shortEncoder(n_channels, L, latent_dims)
        self.lencoder = longEncoder(n_channels, L, latent_dims)
        self.decoder = Decoder(n_channels, L, latent_dims)
        self.in_shape, self.latent_total, self.out_shape = (n_channels, L), 2 * latent_dims, (n_channels,)

    def target_strides(self):
        C, Lw = self.in_shape
        return C * Lw, Lw

    def emit_encoder(self, b, x, mu, logvar):
        lat = self.latent_total // 2
        self.sencoder.emit(b, x, mu.cols(0, lat), logvar.cols(0, lat))
        self.lencoder.emit(b, x, mu.cols(lat, lat), logvar.cols(lat, lat))

    def emit_decoder(self, b, z):
        return self.decoder.emit(b, z)
