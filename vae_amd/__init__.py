"""vae_amd: MI355X-native VAE training hot path (drop-in for AnisKadri/VAE's modules).

Module names mirror the reference's files so `from vae_amd.Encoders import
LongShort_TCVAE_Encoder` replaces `from Encoders import LongShort_TCVAE_Encoder`."""
from . import Decoders, Encoders, mnist_vae, multiscale_vae, optim, parallel, train, utils, vae  # noqa: F401
from ._lib import VaeAmdError, get_compute_dtype, set_compute_dtype  # noqa: F401
from .vae import VariationalAutoencoder  # noqa: F401

__all__ = ["Encoders", "Decoders", "vae", "multiscale_vae", "mnist_vae", "train", "optim", "parallel", "utils", "VariationalAutoencoder",
           "VaeAmdError", "set_compute_dtype", "get_compute_dtype"]
