from .criterion import (
    generator_loss,
    discriminator_loss,
    wasserstein_generator_loss,
    wasserstein_discriminator_loss
)
from .generator import Generator
from .discriminator import Discriminator
from .layers import PixelNorm
