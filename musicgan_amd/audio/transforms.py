"""Input transforms with the reference's semantics (/root/reference/music_gan/audio/transforms.py:4-40); element-wise host-side
tensor expressions that work on whatever device the batch lives on."""
import torch as th


class ChannelMinMaxNorm:
    def __init__(self, epsilon: float = 1e-8):
        self.__epsilon = epsilon

    def __call__(self, x: th.Tensor) -> th.Tensor:
        assert len(x.size()) == 4
        assert x.size()[1] == 2
        flat = x.reshape(x.size()[0], 2, -1)
        x_max = flat.amax(dim=-1).view(-1, 2, 1, 1)
        x_min = flat.amin(dim=-1).view(-1, 2, 1, 1)
        return (x - x_min) / (x_max - x_min + self.__epsilon)


class ChangeRange:
    def __init__(self, lower_bond: float, upper_bound: float):
        self.__range = upper_bound - lower_bond
        self.__start = lower_bond

    def __call__(self, x: th.Tensor) -> th.Tensor:
        return x * self.__range + self.__start
