"""The four loss functions of the reference's interface (names, argument order and values as in
/root/reference/music_gan/networks/criterion.py:4-18).  Their inputs are the (N, 1) critic outputs -- a handful of scalars per
step -- so they are ordinary tensor expressions on whatever device the critic ran on; the critic itself is where the work is."""
import torch


_batch_mean = torch.mean  # the losses are batch means of the (N, 1) critic outputs


def wasserstein_discriminator_loss(y_real: torch.Tensor, y_fake: torch.Tensor) -> torch.Tensor:
    """Critic objective of WGAN: E[D(fake)] - E[D(real)] (the penalty term is added by the caller)."""
    return _batch_mean(y_fake) - _batch_mean(y_real)


def wasserstein_generator_loss(y_fake: torch.Tensor) -> torch.Tensor:
    """Generator objective of WGAN: -E[D(G(z))]."""
    return _batch_mean(y_fake).neg()


def _bits(p: torch.Tensor) -> torch.Tensor:
    return torch.log2(p)


def discriminator_loss(y_real: torch.Tensor, y_fake: torch.Tensor) -> torch.Tensor:
    """Base-2 cross-entropy form for a probability-valued discriminator (kept for interface parity; train() does not use it)."""
    return _batch_mean(_bits(y_real) + _bits(1.0 - y_fake)).neg()


def generator_loss(y_fake: torch.Tensor) -> torch.Tensor:
    """Non-saturating base-2 generator loss for a probability-valued discriminator (interface parity only)."""
    return _batch_mean(_bits(y_fake)).neg()
