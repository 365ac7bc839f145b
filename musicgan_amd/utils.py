"""Host-side bookkeeping of the training driver with the reference's interfaces: the progressive-growing schedule
(`Grower`, /root/reference/music_gan/utils.py:14-86) and the checkpoint writer (`Saver`, :89-242).  Neither is part of the
accelerated path; what is -- the per-batch input transform the schedule selects -- is reached through `transform_batch`."""
from __future__ import annotations

import itertools
import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import audio
from .networks import Discriminator, Generator

_FULL_SIDE = 512        # side of a stored sample
_FIRST_DOWNSCALE = 7    # level 0 trains on 512 / 2**7 = 4 x 4 images


class _InputPipeline:
    """ChannelMinMaxNorm -> ChangeRange(-1, 1) -> Resize(side) as tensor expressions (any device).  Resize stands in for
    torchvision.transforms.Resize(int) on square (N, C, H, W) tensors: bilinear with anti-aliasing, torchvision's tensor
    default (the reference leaves the version unpinned, utils.py:76-80)."""

    def __init__(self, side: int):
        self.side = side
        self._norm = audio.ChannelMinMaxNorm()
        self._range = audio.ChangeRange(-1.0, 1.0)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        y = self._range(self._norm(x))
        if y.shape[-2:] != (self.side, self.side):
            y = F.interpolate(y, size=(self.side, self.side), mode="bilinear", antialias=True, align_corners=False)
        return y


class Grower:
    """Decides when the networks grow and how far the fade-in has progressed, from the number of samples seen.

    `fadein_lengths[k]` samples blend level k in (alpha rises linearly to 1); level k ends once the cumulated
    `train_lengths[:k + 1]` samples have been seen."""

    def __init__(self, n_grow: int, fadein_lengths: List[int], train_lengths: List[int]):
        assert len(fadein_lengths) == n_grow + 1
        assert len(train_lengths) == n_grow
        self._levels = n_grow
        self._fade = list(fadein_lengths)
        self._level_end = list(itertools.accumulate(train_lengths))
        self._state = {"curr_grow": 0, "sample_idx": 0, "step_sample_idx": 0, "downscale": _FIRST_DOWNSCALE}
        self._pipeline = _InputPipeline(self._side())

    def _side(self) -> int:
        return _FULL_SIDE // 2 ** self._state["downscale"]

    # ---- schedule
    def grow(self, viewed_samples: int) -> bool:
        st = self._state
        st["sample_idx"] += viewed_samples
        st["step_sample_idx"] += viewed_samples
        level = st["curr_grow"]
        if level >= self._levels or self._level_end[level] >= st["sample_idx"]:
            return False
        st.update(curr_grow=level + 1, step_sample_idx=0, downscale=st["downscale"] - 1)
        self._pipeline = _InputPipeline(self._side())
        return True

    @property
    def alpha(self) -> float:
        st = self._state
        return min(1.0, (1.0 + st["step_sample_idx"]) / self._fade[st["curr_grow"]])

    @property
    def curr_grow(self) -> int:
        return self._state["curr_grow"]

    # ---- input transform of the current level
    @property
    def scale_transform(self) -> Callable[[torch.Tensor], torch.Tensor]:
        return self._pipeline

    def transform_batch(self, x: torch.Tensor) -> torch.Tensor:
        """`scale_transform(x.to(th.float))` of train.py:139-140.  A batch that already sits on the GPU (float64 as the dataset
        stores it, or float32) goes through the fused kernel (ops.input_transform); a CPU batch takes the tensor expressions."""
        if x.is_cuda:
            from . import ops
            return ops.input_transform(x.contiguous(), self._side())
        return self._pipeline(x.to(torch.float32))

    # ---- resume support (an extension: the reference cannot resume)
    def state_dict(self) -> dict:
        return dict(self._state)

    def load_state_dict(self, sd: dict) -> None:
        self._state = {k: int(sd[k]) for k in ("curr_grow", "sample_idx", "step_sample_idx", "downscale")}
        self._pipeline = _InputPipeline(self._side())


class Saver:
    """Every `save_every` calls of `request_save` writes `{disc,optim_disc,gen,optim_gen}_{k}.pt` (state dicts, the reference's
    file names) into `output_dir`, plus six preview images of fresh samples when matplotlib is importable."""

    _PREVIEWS = 6

    def __init__(self, output_dir: str, save_every: int, rand_channels: int, rand_height: int = 2, rand_width: int = 2):
        self._dir = output_dir
        self._every = save_every
        self._latent = (rand_channels, rand_height, rand_width)
        self._calls = 0
        self._saves = 0

    def _path(self, stem: str, suffix: str = "pt") -> str:
        return os.path.join(self._dir, f"{stem}_{self._saves}.{suffix}")

    def _write_checkpoint(self, gen, disc, optim_gen, optim_disc, train_state: Optional[Callable[[], dict]]) -> None:
        for stem, obj in (("disc", disc), ("optim_disc", optim_disc), ("gen", gen), ("optim_gen", optim_gen)):
            torch.save(obj.state_dict(), self._path(stem))
        if train_state is not None:
            torch.save(train_state(), self._path("train_state"))

    def _write_previews(self, gen: Generator, alpha: float) -> None:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            return  # previews are optional: matplotlib is not a dependency of the accelerated package
        device = next(gen.parameters()).device
        with torch.no_grad():
            for sample in range(self._PREVIEWS):
                x_fake = gen(torch.randn(1, *self._latent, device=device), alpha)[0].cpu().numpy()
                for channel, name in enumerate(("magn", "phase")):
                    img = x_fake[channel]
                    fig, ax = plt.subplots()
                    ax.matshow(img / (img.max() - img.min()), cmap="plasma")
                    plt.title(f"gen {name} {self._saves} grow={gen.curr_layer}")
                    fig.savefig(os.path.join(self._dir, f"{name}_{self._saves}_ID{sample}.png"))
                    plt.close()

    def request_save(self, gen: Generator, disc: Discriminator, optim_gen, optim_disc, alpha: float,
                     train_state: Optional[Callable[[], dict]] = None) -> bool:
        """`train_state` (optional callable -> dict) is an extension over the reference: the growth level, the Grower counters
        and the iteration index are written next to the four reference files as `train_state_{k}.pt`, which is what
        `train(..., resume_from=...)` needs (the reference saves weights only, utils.py:118-145)."""
        self._calls += 1
        if self._calls % self._every:
            return False
        self._write_checkpoint(gen, disc, optim_gen, optim_disc, train_state)
        self._write_previews(gen, alpha)
        self._saves += 1
        return True

    # ---- extensions for data-parallel runs and resume (the reference has neither)
    def due(self) -> bool:
        """Whether the next `request_save` call writes a checkpoint."""
        return (self._calls + 1) % self._every == 0

    def tick(self) -> None:
        """Count one call without writing (ranks other than 0 keep their cadence in step with rank 0)."""
        self._calls += 1
        if self._calls % self._every == 0:
            self._saves += 1

    def state_dict(self) -> dict:
        return {"calls": self._calls, "saves": self._saves}

    def state_dict_after_save(self) -> dict:
        """The counters as they stand once the checkpoint being written is complete (for the `train_state` written with it)."""
        return {"calls": self._calls, "saves": self._saves + 1}

    def load_state_dict(self, sd: dict) -> None:
        self._calls, self._saves = int(sd["calls"]), int(sd["saves"])

    @property
    def curr_save(self) -> int:
        return self._saves - 1  # index of the last checkpoint written

    @property
    def save_counter(self) -> int:
        return self._calls % self._every
