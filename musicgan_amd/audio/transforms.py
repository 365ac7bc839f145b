"""The two element-wise transforms of the input pipeline, interface of /root/reference/music_gan/audio/transforms.py:4-40.
Plain tensor expressions that run wherever the batch lives; the training loop itself goes through the fused device kernel
(`Grower.transform_batch` -> ops.input_transform), which evaluates the same formulas."""
import torch


class ChannelMinMaxNorm:
    """(N, 2, H, W) -> each channel of each item mapped to [0, 1): (x - min) / (max - min + epsilon)."""

    def __init__(self, epsilon: float = 1e-8):
        self._eps = float(epsilon)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 4, f"expected (N, 2, H, W), got {tuple(x.shape)}"
        assert x.shape[1] == 2, f"expected 2 channels (magnitude, phase), got {x.shape[1]}"
        lo, hi = torch.aminmax(x.flatten(2), dim=2)
        lo, hi = lo[..., None, None], hi[..., None, None]
        return (x - lo) / (hi - lo + self._eps)


class ChangeRange:
    """[0, 1] -> [lower_bond, upper_bound] (the reference's argument spelling is part of the interface)."""

    def __init__(self, lower_bond: float, upper_bound: float):
        self._offset = lower_bond
        self._span = upper_bound - lower_bond

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return x * self._span + self._offset
