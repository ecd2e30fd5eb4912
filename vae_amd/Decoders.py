"""MI355X drop-ins for the reference's conv decoders (Decoders.py:16-78).

RnnDecoder (Decoders.py:83-126) is outside the hot path.
"""
import torch.nn as nn

from . import _lib as L
from .Encoders import lin_size  # same function in the reference (Decoders.py:16-28)
from .module import BatchNormParams, ConvParams, HipModule, LinearParams, NoParams


class TCVAE_Decoder(HipModule):
    """Linear(latent -> 2CN) viewed as (2CN, 1), then N x [ConvTranspose1d k2 s2 -> LeakyReLU
    (-> BatchNorm1d except after the last)], output = time index 0 (Decoders.py:35-65)."""

    def __init__(self, input_size, num_layers, latent_dims, L=30, slope=0.2, first_kernel=None):
        super().__init__()
        self.input_size, self.num_layers, self.latent_dims, self.slope = input_size, num_layers, latent_dims, float(slope)
        self.n = lin_size(L, num_layers, first_kernel)
        self.decoder_lin = LinearParams(latent_dims, input_size * 2 * num_layers)
        self.cnn_layers = nn.ModuleList()
        for i in range(num_layers, 0, -1):
            cin = input_size * 2 * i
            cout = input_size if i == 1 else input_size * 2 * (i - 1)
            # ConvTranspose1d weight is (in, out, k); torch's fan_in for it is out*k
            self.cnn_layers.extend([ConvParams((cin, cout, 2), cout, cout * 2), NoParams()])
            if i != 1:
                self.cnn_layers.append(BatchNormParams(cout))

    def emit(self, b, z):
        C, N = self.input_size, self.num_layers
        h = b.linear(z, self.decoder_lin.weight, self.decoder_lin.bias, self.latent_dims, 2 * C * N)
        length = 1
        for j in range(N):
            conv = self.cnn_layers[3 * j]
            cin, cout, _ = conv.weight.shape
            h, (_, length) = b.conv(h, conv.weight, conv.bias, cin, (1, length), cout, (1, 2), (1, 2), (0, 0),
                                    transposed=True, act=L.ACT_LEAKY, slope=self.slope)
            if j != N - 1:
                h = b.batchnorm(h, self.cnn_layers[3 * j + 2], cout, length)
        return b.slice0(h, C, length)

    def emit_program(self, b, training):
        z = b.input("z", self.latent_dims, grad=True)
        b.output("x_rec", self.emit(b, z), (self.input_size,))

    def forward(self, z):
        return self.run_program([z])["x_rec"]


class LongShort_TCVAE_Decoder(HipModule):
    """TCVAE_Decoder on the doubled latent (Decoders.py:71-78)."""

    def __init__(self, input_size, hidden_size, num_layers, latent_dims, L=30, slope=0.2, first_kernel=None):
        super().__init__()
        self.longshort_decoder = TCVAE_Decoder(input_size, num_layers, 2 * latent_dims, L, slope, first_kernel)
        self.input_size, self.latent_dims = input_size, 2 * latent_dims

    def emit(self, b, z):
        return self.longshort_decoder.emit(b, z)

    def emit_program(self, b, training):
        z = b.input("z", self.latent_dims, grad=True)
        b.output("x_rec", self.emit(b, z), (self.input_size,))

    def forward(self, z):
        return self.run_program([z])["x_rec"]
