"""MI355X drop-in for the notebook-only MNIST Conv2d VAE
(.ipynb_checkpoints/vae-checkpoint.ipynb:31-130) and its BCE ELBO (:141-151).

`channels` / `hidden` are additive knobs for BASELINE.json's "widened conv VAE" config
(4x channels, latent 256); defaults reproduce the notebook model (103,343 parameters)."""
import torch.nn as nn

from . import _lib as L
from .module import BatchNormParams, ConvParams, HipModule, LinearParams, NoParams
from .vae import HipVAE


class VariationalEncoder(HipModule):
    """Conv2d(1,c0,3,s2,p1) ReLU Conv2d(c0,c1,3,s2,p1) BN ReLU Conv2d(c1,c2,3,s2,p0) ReLU, then two
    heads Linear(9*c2, hidden) ReLU Linear(hidden, latent) (notebook :31-69)."""

    def __init__(self, latent_dims, channels=(8, 16, 32), hidden=100):
        super().__init__()
        c0, c1, c2 = channels
        self.channels, self.hidden, self.latent_dims = tuple(channels), hidden, latent_dims
        self.encoder_cnn = nn.ModuleList([
            ConvParams((c0, 1, 3, 3), c0, 9), NoParams(),
            ConvParams((c1, c0, 3, 3), c1, c0 * 9), BatchNormParams(c1), NoParams(),
            ConvParams((c2, c1, 3, 3), c2, c1 * 9), NoParams()])
        self.encoder_mean = nn.ModuleList([LinearParams(9 * c2, hidden), NoParams(), LinearParams(hidden, latent_dims)])
        self.encoder_std = nn.ModuleList([LinearParams(9 * c2, hidden), NoParams(), LinearParams(hidden, latent_dims)])

    def emit(self, b, x, mu_out, logvar_out):
        c0, c1, c2 = self.channels
        cnn = self.encoder_cnn
        h, hw = b.conv(x, cnn[0].weight, cnn[0].bias, 1, (28, 28), c0, (3, 3), (2, 2), (1, 1), act=L.ACT_LEAKY)
        h, hw = b.conv(h, cnn[2].weight, cnn[2].bias, c0, hw, c1, (3, 3), (2, 2), (1, 1))
        h = b.batchnorm(h, cnn[3], c1, hw[0] * hw[1], relu=True)
        h, hw = b.conv(h, cnn[5].weight, cnn[5].bias, c1, hw, c2, (3, 3), (2, 2), (0, 0), act=L.ACT_LEAKY)
        K = c2 * hw[0] * hw[1]
        # both branches read h: the std branch (emitted second) runs first in backward and stores d(h); their first
        # layers are emitted next to each other so that the executor runs the pair as one launch (forward and d(h))
        tm = b.linear(h, self.encoder_mean[0].weight, self.encoder_mean[0].bias, K, self.hidden, act=L.ACT_LEAKY,
                      acc_din=True)
        ts = b.linear(h, self.encoder_std[0].weight, self.encoder_std[0].bias, K, self.hidden, act=L.ACT_LEAKY)
        b.linear(tm, self.encoder_mean[2].weight, self.encoder_mean[2].bias, self.hidden, self.latent_dims, out=mu_out)
        b.linear(ts, self.encoder_std[2].weight, self.encoder_std[2].bias, self.hidden, self.latent_dims, out=logvar_out)

    def emit_program(self, b, training):
        x = b.input("x", 784)
        mu, logvar = b.act(self.latent_dims), b.act(self.latent_dims)
        self.emit(b, x, mu, logvar)
        b.output("mu", mu, (self.latent_dims,))
        b.output("sigma", logvar, (self.latent_dims,))

    def forward(self, x):
        out = self.run_program([x])
        return out["mu"], out["sigma"]


class Decoder(HipModule):
    """Linear(latent, hidden) ReLU Linear(hidden, 9*c2) -> (c2,3,3) -> ConvT(c2,c1,3,s2) BN ReLU
    ConvT(c1,c0,3,s2,p1,op1) BN ReLU ConvT(c0,1,3,s2,p1,op1) sigmoid (notebook :78-107)."""

    def __init__(self, latent_dims, channels=(8, 16, 32), hidden=100):
        super().__init__()
        c0, c1, c2 = channels
        self.channels, self.hidden, self.latent_dims = tuple(channels), hidden, latent_dims
        self.decoder_lin = nn.ModuleList([LinearParams(latent_dims, hidden), NoParams(), LinearParams(hidden, 9 * c2)])
        self.decoder_conv = nn.ModuleList([
            ConvParams((c2, c1, 3, 3), c1, c1 * 9), BatchNormParams(c1), NoParams(),
            ConvParams((c1, c0, 3, 3), c0, c0 * 9), BatchNormParams(c0), NoParams(),
            ConvParams((c0, 1, 3, 3), 1, 9)])

    def emit(self, b, z):
        c0, c1, c2 = self.channels
        lin, cv = self.decoder_lin, self.decoder_conv
        h = b.linear(z, lin[0].weight, lin[0].bias, self.latent_dims, self.hidden, act=L.ACT_LEAKY)
        h = b.linear(h, lin[2].weight, lin[2].bias, self.hidden, 9 * c2)
        h, hw = b.conv(h, cv[0].weight, cv[0].bias, c2, (3, 3), c1, (3, 3), (2, 2), (0, 0), transposed=True)
        h = b.batchnorm(h, cv[1], c1, hw[0] * hw[1], relu=True)
        h, hw = b.conv(h, cv[3].weight, cv[3].bias, c1, hw, c0, (3, 3), (2, 2), (1, 1), transposed=True, out_pad=(1, 1))
        h = b.batchnorm(h, cv[4], c0, hw[0] * hw[1], relu=True)
        h, hw = b.conv(h, cv[6].weight, cv[6].bias, c0, hw, 1, (3, 3), (2, 2), (1, 1), transposed=True,
                       out_pad=(1, 1), act=L.ACT_SIGMOID)
        assert hw == (28, 28)
        return h

    def emit_program(self, b, training):
        z = b.input("z", self.latent_dims, grad=True)
        b.output("x_rec", self.emit(b, z), (1, 28, 28))

    def forward(self, z):
        return self.run_program([z])["x_rec"]


class VariationalAutoencoder(HipVAE):
    """Notebook :115-130.  forward(x) -> (x_rec (B,1,28,28) in (0,1), mu, logvar)."""

    elbo_bce = True

    def __init__(self, latent_dims, channels=(8, 16, 32), hidden=100):
        super().__init__()
        self.encoder = VariationalEncoder(latent_dims, channels, hidden)
        self.decoder = Decoder(latent_dims, channels, hidden)
        self.in_shape, self.latent_total, self.out_shape = (1, 28, 28), latent_dims, (1, 28, 28)

    def target_strides(self):
        return 784, 1  # ELBO target is the image itself (notebook :163)

    def emit_encoder(self, b, x, mu, logvar):
        self.encoder.emit(b, x, mu, logvar)

    def emit_decoder(self, b, z):
        return self.decoder.emit(b, z)
