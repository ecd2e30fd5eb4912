"""MI355X drop-ins for the reference's conv encoders (Encoders.py).

Same class names, constructor signatures, attribute names and state_dict keys as the
reference (Encoders.py:16-107); forward runs as a libvae_amd program (one C call).
RnnEncoder (Encoders.py:110-147) is outside the hot path.
"""
import torch.nn as nn

from . import _lib as L
from .module import BatchNormParams, ConvParams, HipModule, LinearParams, NoParams


def lin_size(n, num_layers, first_kernel=None):
    """Flatten-size helper with the reference's semantics (Encoders.py:16-28): length after
    num_layers stride-2 convolutions (kernel first_kernel in layer 0, else 2) times 2*num_layers."""
    for layer in range(num_layers):
        k = first_kernel if (layer == 0 and first_kernel is not None) else 2
        n = 1 + (n - k) // 2
    if n <= 0:
        raise ValueError("Window Length is too small in relation to the number of Layers")
    return n * 2 * num_layers


class TCVAE_Encoder(HipModule):
    """N x [Conv1d stride 2 -> LeakyReLU(slope) -> BatchNorm1d] -> flatten -> mu / logvar heads
    (Encoders.py:31-87).  Channels grow as 2*C*(i+1)."""

    def __init__(self, input_size, num_layers, latent_dims, L=30, slope=0.2, first_kernel=None):
        super().__init__()
        self.n = lin_size(L, num_layers, first_kernel)
        self.input_size, self.num_layers, self.latent_dims = input_size, num_layers, latent_dims
        self.L, self.slope = L, float(slope)
        self.first_kernel = 2 if first_kernel is None else first_kernel
        self.cnn_layers = nn.ModuleList()
        for i in range(num_layers):
            cin = input_size if i == 0 else input_size * 2 * i
            k = self.first_kernel if i == 0 else 2
            self.cnn_layers.extend([ConvParams((input_size * 2 * (i + 1), cin, k), input_size * 2 * (i + 1), cin * k),
                                    NoParams(), BatchNormParams(input_size * 2 * (i + 1))])
        self.encoder_mu = nn.ModuleList([LinearParams(self.n * input_size, latent_dims)])
        self.encoder_logvar = nn.ModuleList([LinearParams(self.n * input_size, latent_dims)])

    def emit(self, b, x, mu_out, logvar_out):
        C, h, length = self.input_size, x, self.L
        for i in range(self.num_layers):
            conv, bn = self.cnn_layers[3 * i], self.cnn_layers[3 * i + 2]
            cout, cin, k = conv.weight.shape
            h, (_, length) = b.conv(h, conv.weight, conv.bias, cin, (1, length), cout, (1, k), (1, 2), (0, 0),
                                    act=L.ACT_LEAKY, slope=self.slope)
            h = b.batchnorm(h, bn, cout, length)
        K = self.n * C
        # h feeds two heads: in backward the logvar head (emitted last) runs first and stores
        # d(h); the mu head then accumulates into it.
        b.linear(h, self.encoder_mu[0].weight, self.encoder_mu[0].bias, K, self.latent_dims, out=mu_out, acc_din=True)
        b.linear(h, self.encoder_logvar[0].weight, self.encoder_logvar[0].bias, K, self.latent_dims, out=logvar_out)

    def emit_program(self, b, training):
        x = b.input("x", self.input_size * self.L)
        mu, logvar = b.act(self.latent_dims), b.act(self.latent_dims)
        self.emit(b, x, mu, logvar)
        b.output("mu", mu, (self.latent_dims,))
        b.output("logvar", logvar, (self.latent_dims,))

    def forward(self, x):
        out = self.run_program([x])
        return out["mu"], out["logvar"]


class LongShort_TCVAE_Encoder(HipModule):
    """Short (first kernel 2) and long (first kernel `first_kernel`) encoders on the same window;
    mu / logvar are the concatenations [short | long] (Encoders.py:93-107).  The heads write
    straight into the two halves of one buffer, so there is no cat kernel."""

    def __init__(self, input_size, hidden_size, num_layers, latent_dims, L=30, slope=0.2, first_kernel=None):
        super().__init__()
        self.short_encoder = TCVAE_Encoder(input_size, num_layers, latent_dims, L, slope, first_kernel=None)
        self.long_encoder = TCVAE_Encoder(input_size, num_layers, latent_dims, L, slope, first_kernel)
        self.input_size, self.L, self.latent_total = input_size, L, 2 * latent_dims

    def emit(self, b, x, mu, logvar):
        lat = self.latent_total // 2
        self.short_encoder.emit(b, x, mu.cols(0, lat), logvar.cols(0, lat))
        self.long_encoder.emit(b, x, mu.cols(lat, lat), logvar.cols(lat, lat))

    def emit_program(self, b, training):
        x = b.input("x", self.input_size * self.L)
        mu, logvar = b.act(self.latent_total), b.act(self.latent_total)
        self.emit(b, x, mu, logvar)
        b.output("mu", mu, (self.latent_total,))
        b.output("logvar", logvar, (self.latent_total,))

    def forward(self, x):
        out = self.run_program([x])
        return out["mu"], out["logvar"]
