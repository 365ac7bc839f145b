"""Progressive-growing schedule and checkpoint writer with the reference's interfaces
(/root/reference/music_gan/utils.py:14-86 Grower, :89-242 Saver).  Host-side bookkeeping, not part of the accelerated path."""
from __future__ import annotations

from os.path import join
from typing import List

import torch as th
import torch.nn.functional as F

from . import audio
from .networks import Discriminator, Generator


class _Resize:
    """Stand-in for torchvision.transforms.Resize(int) on (N,C,H,W) square tensors (utils.py:76-80): bilinear with
    anti-aliasing (torchvision's current tensor default; the reference leaves the version unpinned)."""

    def __init__(self, size: int):
        self.size = size

    def __call__(self, x: th.Tensor) -> th.Tensor:
        if x.shape[-1] == self.size and x.shape[-2] == self.size:
            return x
        return F.interpolate(x, size=(self.size, self.size), mode="bilinear", antialias=True, align_corners=False)


class _Compose:
    def __init__(self, fns):
        self.fns = fns

    def __call__(self, x):
        for f in self.fns:
            x = f(x)
        return x


class Grower:
    def __init__(self, n_grow: int, fadein_lengths: List[int], train_lengths: List[int]):
        self.__curr_grow = 0
        self.__n_grow = n_grow
        self.__sample_idx = 0
        self.__step_sample_idx = 0
        self.__downscale = 7
        self.__transform = Grower.__get_transform(self.__downscale)
        assert len(fadein_lengths) == self.__n_grow + 1
        assert len(train_lengths) == self.__n_grow
        self.__fadein_l = fadein_lengths
        acc, cum = 0, []
        for t in train_lengths:
            acc += t
            cum.append(acc)
        self.__train_l = cum

    def grow(self, viewed_samples: int) -> bool:
        self.__sample_idx += viewed_samples
        self.__step_sample_idx += viewed_samples
        if self.__curr_grow >= self.__n_grow:
            return False
        if self.__train_l[self.__curr_grow] < self.__sample_idx:
            self.__step_sample_idx = 0
            self.__curr_grow += 1
            self.__downscale -= 1
            self.__transform = Grower.__get_transform(self.__downscale)
            return True
        return False

    @property
    def alpha(self) -> float:
        return min(1., (1. + self.__step_sample_idx) / self.__fadein_l[self.__curr_grow])

    @property
    def curr_grow(self) -> int:
        return self.__curr_grow

    @staticmethod
    def __get_transform(downscale_factor: int) -> _Compose:
        target_size = 512 // 2 ** downscale_factor
        return _Compose([audio.ChannelMinMaxNorm(), audio.ChangeRange(-1., 1.), _Resize(target_size)])

    @property
    def scale_transform(self) -> _Compose:
        return self.__transform

    def transform_batch(self, x: th.Tensor) -> th.Tensor:
        """`scale_transform(x.to(th.float))` of train.py:139-140.  A batch that already sits on the GPU (float64 as the dataset
        stores it, or float32) goes through the fused kernel (ops.input_transform); a CPU batch takes the tensor expressions."""
        if x.is_cuda:
            from . import ops
            return ops.input_transform(x.contiguous(), 512 // 2 ** self.__downscale)
        return self.__transform(x.to(th.float))

    def state_dict(self):
        return {"curr_grow": self.__curr_grow, "sample_idx": self.__sample_idx,
                "step_sample_idx": self.__step_sample_idx, "downscale": self.__downscale}

    def load_state_dict(self, sd):
        self.__curr_grow, self.__sample_idx = sd["curr_grow"], sd["sample_idx"]
        self.__step_sample_idx, self.__downscale = sd["step_sample_idx"], sd["downscale"]
        self.__transform = Grower.__get_transform(self.__downscale)


class Saver:
    def __init__(self, output_dir: str, save_every: int, rand_channels: int, rand_height: int = 2, rand_width: int = 2):
        self.__output_dir = output_dir
        self.__counter = 0
        self.__curr_save = 0
        self.__save_every = save_every
        self.__rand_channels = rand_channels
        self.__height = rand_height
        self.__width = rand_width
        self.__nb_output_images = 6

    def __save_models(self, gen: Generator, disc: Discriminator, optim_gen, optim_disc):
        th.save(disc.state_dict(), join(self.__output_dir, f"disc_{self.__curr_save}.pt"))
        th.save(optim_disc.state_dict(), join(self.__output_dir, f"optim_disc_{self.__curr_save}.pt"))
        th.save(gen.state_dict(), join(self.__output_dir, f"gen_{self.__curr_save}.pt"))
        th.save(optim_gen.state_dict(), join(self.__output_dir, f"optim_gen_{self.__curr_save}.pt"))

    def __save_outputs(self, gen: Generator, alpha: float):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            return  # previews are optional: matplotlib is not a dependency of the accelerated package
        device = next(gen.parameters()).device
        with th.no_grad():
            for gen_idx in range(self.__nb_output_images):
                z = th.randn(1, self.__rand_channels, self.__height, self.__width, device=device)
                x_fake = gen(z, alpha)
                for name, ch in (("magn", 0), ("phase", 1)):
                    img = x_fake[0, ch].detach().cpu().numpy()
                    fig, ax = plt.subplots()
                    ax.matshow(img / (img.max() - img.min()), cmap="plasma")
                    plt.title(f"gen {name} {self.__curr_save} grow={gen.curr_layer}")
                    fig.savefig(join(self.__output_dir, f"{name}_{self.__curr_save}_ID{gen_idx}.png"))
                    plt.close()

    def request_save(self, gen: Generator, disc: Discriminator, optim_gen, optim_disc, alpha: float,
                     train_state=None) -> bool:
        """`train_state` (optional callable -> dict) is an extension over the reference: the growth level, the Grower counters
        and the iteration index are written next to the four reference files as `train_state_{k}.pt`, which is what
        `train(..., resume_from=...)` needs (the reference cannot resume: utils.py:118-145 saves weights only)."""
        self.__counter += 1
        if self.__counter % self.__save_every == 0:
            self.__save_models(gen, disc, optim_gen, optim_disc)
            if train_state is not None:
                th.save(train_state(), join(self.__output_dir, f"train_state_{self.__curr_save}.pt"))
            self.__save_outputs(gen, alpha)
            self.__curr_save += 1
            return True
        return False

    @property
    def curr_save(self) -> int:
        return self.__curr_save - 1

    @property
    def save_counter(self) -> int:
        return self.__counter % self.__save_every
