"""Public surface of the reference's `networks` package (/root/reference/music_gan/networks/__init__.py): the two modules,
PixelNorm and the four losses."""
from . import criterion as _criterion
from .discriminator import Discriminator
from .generator import Generator
from .layers import PixelNorm

wasserstein_discriminator_loss = _criterion.wasserstein_discriminator_loss
wasserstein_generator_loss = _criterion.wasserstein_generator_loss
discriminator_loss = _criterion.discriminator_loss
generator_loss = _criterion.generator_loss

__all__ = ["Generator", "Discriminator", "PixelNorm", "wasserstein_discriminator_loss", "wasserstein_generator_loss",
           "discriminator_loss", "generator_loss"]
