"""`train(run_name, input_dataset_path, output_dir)` with the reference's signature, hyper-parameters, growth schedule, D:G
cadence and checkpoint cadence (/root/reference/music_gan/train.py:18-278) on the MI355X stepper.

Differences, all outside the arithmetic: mlflow is optional; the six per-iteration metrics stay on the device in a window buffer
and come back with ONE copy every `metric_every` iterations (the reference does 4-6 `.item()` syncs per iteration,
train.py:180-186,218-221); launched under torchrun (WORLD_SIZE > 1) it runs data-parallel, one process per GPU, each rank on
its own shard of a per-epoch permutation and with its own latent / epsilon noise stream, the growth schedule advanced by the
GLOBAL sample count so that all ranks grow on the same iteration; and a run can be resumed (`resume_from`) bit-identically --
the reference can only save (train.py never loads)."""
from __future__ import annotations

import glob
import os
import re
from os import mkdir
from os.path import exists, isdir
from statistics import mean
from typing import Dict, Iterator, List, Optional

import torch as th
from torch.utils.data import DataLoader, Sampler

from . import audio, networks
from .optim import FusedAdam
from .train_step import ProGANStepper
from .utils import Grower, Saver

try:  # optional, exactly the calls the reference makes
    import mlflow
except ImportError:  # pragma: no cover
    mlflow = None

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    def tqdm(x, **_):
        return x

_METRICS = ("disc_loss", "grad_pen", "e_tp", "e_tn", "gen_loss", "e_gen")


class ShardedShuffle(Sampler):
    """The reference's `shuffle=True, drop_last=True` loader order (train.py:77-84) made reproducible and shardable: epoch e
    visits `randperm(n, seed + e)`; rank r of w takes every w-th index starting at r (so the union over ranks is one
    permutation and `w` ranks at batch B see what one rank at batch w*B would); `skip` drops the first indices of the epoch
    (resume in mid-epoch)."""

    def __init__(self, n: int, seed: int, rank: int = 0, world: int = 1):
        self.n, self.seed, self.rank, self.world = n, seed, rank, world
        self.epoch = self.skip = 0

    def set_epoch(self, epoch: int, skip: int = 0) -> None:
        self.epoch, self.skip = epoch, skip

    def _indices(self) -> List[int]:
        g = th.Generator().manual_seed(self.seed + self.epoch)
        perm = th.randperm(self.n, generator=g).tolist()
        per_rank = self.n // self.world
        return perm[self.rank:per_rank * self.world:self.world]

    def __iter__(self) -> Iterator[int]:
        return iter(self._indices()[self.skip:])

    def __len__(self) -> int:
        return max(0, self.n // self.world - self.skip)


class MetricWindow:
    """Device-side window of the per-iteration scalars: `push` writes one row with a single tiny kernel (no host sync),
    `flush` brings all rows back with ONE device-to-host copy and feeds the reference's 20-entry sliding windows
    (train.py:120-127,177-186,216-221)."""

    def __init__(self, capacity: int, device, window: int = 20):
        self._buf = th.full((capacity, len(_METRICS)), float("nan"), dtype=th.float32, device=device)
        self._rows = 0
        self.hist: Dict[str, List[float]] = {k: [0.] * window for k in _METRICS}
        self.last: Dict[str, float] = {k: 0. for k in _METRICS}

    def push(self, d: Dict[str, th.Tensor], g: Optional[Dict[str, th.Tensor]]) -> None:
        if self._rows == self._buf.shape[0]:
            self.flush()
        nan = self._buf.new_full((), float("nan"))
        row = [d["disc_loss"], d["grad_pen"], d["out_real_mean"], d["out_fake_mean"],
               g["gen_loss"] if g is not None else nan, g["out_fake_mean"] if g is not None else nan]
        th.stack([r.reshape(()) for r in row], out=self._buf[self._rows])
        self._rows += 1

    def flush(self) -> None:
        if self._rows == 0:
            return
        host = self._buf[:self._rows].cpu()  # the only device->host copy of the window
        self._rows = 0
        for row in host.tolist():
            for key, v in zip(_METRICS, row):
                if v == v:  # generator columns are NaN on iterations without a generator step
                    self.hist[key] = self.hist[key][1:] + [v]
                    self.last[key] = v


def _latest_state(resume_from: str) -> str:
    """Newest `train_state_k.pt` by the iteration it was taken at (k restarts at 0 in a fresh output directory)."""
    states = glob.glob(os.path.join(resume_from, "train_state_*.pt"))
    assert states, f"no train_state_*.pt in \"{resume_from}\""
    best = max(states, key=lambda p: (int(th.load(p)["iter_idx"]), int(re.search(r"_(\d+)\.pt$", p).group(1))))
    return re.search(r"train_state_(\d+)\.pt$", best).group(1)


def train(run_name: str, input_dataset_path: str, output_dir: str, *, nb_epoch: int = 1000, batch_size: int = 6,
          num_workers: int = 6, metric_every: int = 20, max_iters: int = 0, save_every: int = 1000,
          resume_from: str = None, fadein_lengths=None, train_lengths=None, rand_channels: int = 32,
          use_packed_loader: bool = True, progress_hook=None) -> None:
    """Reference signature plus keyword-only extensions (all defaulting to the reference's literals).  `resume_from`: a
    directory written by a previous run; its newest `train_state_k.pt` / `gen_k.pt` / `disc_k.pt` / `optim_*_k.pt` set is
    loaded (growth level, Grower counters, weights, Adam state, noise stream, position in the epoch, checkpoint numbering), after
    which the run continues exactly as the uninterrupted one would (tests/test_audio_gpu.py::test_resume_is_bit_identical).
    `progress_hook(iter_idx)`: called at the end of every iteration (bench.py's `train_loop` record takes its time stamps there)."""
    assert isdir(input_dataset_path), f"\"{input_dataset_path}\" doesn't exist or is not a directory"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # dmabuf IPC (RCCL between processes needs it on this image), in this process and before the first HIP call: a run started as
        # `torchrun ... -m musicgan_amd train` has no launcher of ours in front of it
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    th.cuda.set_device(local_rank)
    device = th.device("cuda", local_rank)
    if world > 1 and not th.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        th.distributed.init_process_group(backend=os.environ.get("MG_DIST_BACKEND", "nccl"), device_id=device)
    if world > 1:
        from .dist import check_world
        check_world(world, device)  # every rank takes part in the data path's collectives, or the run ends here (non-zero exit)
    if mlflow is not None and rank == 0:
        mlflow.set_experiment("music_gan")
        mlflow.start_run(run_name=run_name)
        mlflow.start_run(run_name="train", nested=True)

    sample_rate = audio.SAMPLE_RATE
    height, width = 2, 2
    disc_lr = gen_lr = 1e-3
    betas = (0.0, 0.9)

    if not exists(output_dir):
        if rank == 0:
            mkdir(output_dir)
    elif not isdir(output_dir):
        raise NotADirectoryError(f"\"{output_dir}\" is not a directory !")

    resume_state, k = None, None
    if resume_from is not None:
        k = _latest_state(resume_from)
        resume_state = th.load(os.path.join(resume_from, f"train_state_{k}.pt"))
    # one seed for everything shared by the ranks (initial weights, fresh heads / stems at growth, the epoch permutations);
    # the latent and epsilon noise comes from a per-rank stream, so ranks draw DIFFERENT fake / interpolated batches
    base_seed = int(resume_state["base_seed"]) if resume_state else (0 if world > 1 else th.initial_seed() % (2 ** 31))
    th.manual_seed(base_seed)
    gen = networks.Generator(rand_channels, end_layer=0).to(device)
    disc = networks.Discriminator(start_layer=7).to(device)
    from .dist import broadcast_parameters
    broadcast_parameters([gen, disc])
    optim_gen = FusedAdam(gen.parameters(), lr=gen_lr, betas=betas)
    optim_disc = FusedAdam(disc.parameters(), lr=disc_lr, betas=betas)
    noise = th.Generator(device=device)
    noise.manual_seed(base_seed + 7919 * (rank + 1))

    def grow_networks() -> None:  # train.py:258-272
        th.manual_seed(base_seed + 1000 + gen.curr_layer)  # identical new head/stem on every rank (and after a resume)
        gen.next_layer()
        disc.next_layer()
        optim_gen.add_param_group({"params": gen.end_block_params(), "lr": gen_lr, "betas": betas})
        optim_disc.add_param_group({"params": disc.start_block_parameters(), "lr": disc_lr, "betas": betas})

    start_iter = start_epoch = epoch_pos = 0
    if resume_state is not None:
        for _ in range(resume_state["level"]):  # replay the growth so parameter groups line up with the saved optimizers
            grow_networks()
        gen.load_state_dict(th.load(os.path.join(resume_from, f"gen_{k}.pt"), map_location=device))
        disc.load_state_dict(th.load(os.path.join(resume_from, f"disc_{k}.pt"), map_location=device))
        optim_gen.load_state_dict(th.load(os.path.join(resume_from, f"optim_gen_{k}.pt"), map_location=device))
        optim_disc.load_state_dict(th.load(os.path.join(resume_from, f"optim_disc_{k}.pt"), map_location=device))
        for opt in (optim_gen, optim_disc):  # torch keeps `step` on the host, the moments on the device
            for st in opt.state.values():
                st["step"] = st["step"].to("cpu")
        start_iter, start_epoch = int(resume_state["iter_idx"]), int(resume_state["epoch"])
        epoch_pos = int(resume_state["epoch_pos"])
        noise.set_state(resume_state["noise_rng"][rank])
    stepper = ProGANStepper(gen, disc, optim_gen, optim_disc, rand_channels, height, width, noise=noise)

    # the float32 memory-mapped side-car when the dataset has one (one gather + one asynchronous upload per batch on a background
    # thread), the reference's per-sample th.load through DataLoader workers otherwise; same samples in the same order either way
    use_packed = use_packed_loader and audio.has_packed(input_dataset_path)
    audio_dataset = audio.PackedAudioDataset(input_dataset_path) if use_packed else audio.AudioDataset(input_dataset_path)
    sampler = ShardedShuffle(len(audio_dataset), base_seed, rank, world)
    if use_packed:
        data_loader = audio.PackedLoader(audio_dataset, batch_size, sampler, device)
    else:
        data_loader = DataLoader(audio_dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                                 drop_last=True, pin_memory=True)

    if mlflow is not None and rank == 0:
        mlflow.log_params({"input_dataset": input_dataset_path, "nb_sample": len(audio_dataset),
                           "output_dir": output_dir, "rand_channels": rand_channels, "nb_epoch": nb_epoch,
                           "batch_size": batch_size, "disc_lr": disc_lr, "gen_lr": gen_lr, "betas": betas,
                           "sample_rate": sample_rate, "width": width, "height": height, "world_size": world})

    grower = Grower(n_grow=7, fadein_lengths=fadein_lengths or [1, 25000, 37500, 50000, 62500, 75000, 87500, 100000],
                    train_lengths=train_lengths or [50000, 100000, 150000, 200000, 250000, 300000, 350000])
    saver = Saver(output_dir, save_every=save_every, rand_channels=rand_channels, rand_height=height, rand_width=width)
    if resume_state is not None:
        grower.load_state_dict(resume_state["grower"])
        same_dir = os.path.realpath(resume_from) == os.path.realpath(output_dir)
        # continuing in place: keep numbering, so nothing already written is overwritten; a fresh directory starts at 0 again
        saver.load_state_dict(resume_state["saver"] if same_dir else {**resume_state["saver"], "saves": 0})

    def noise_states() -> List[th.Tensor]:
        mine = noise.get_state()
        if world == 1:
            return [mine]
        mine = mine.to(device)
        bufs = [th.empty_like(mine) for _ in range(world)] if rank == 0 else None
        th.distributed.gather(mine, bufs, dst=0)
        return [b.cpu() for b in bufs] if rank == 0 else []

    metrics = MetricWindow(metric_every, device)
    iter_idx = start_iter
    for e in range(start_epoch, nb_epoch):
        sampler.set_epoch(e, skip=epoch_pos * batch_size if e == start_epoch else 0)
        pos = epoch_pos if e == start_epoch else 0
        bar = tqdm(data_loader) if rank == 0 else data_loader
        for x_real in bar:
            # float64 -> float32, per-channel min-max to [-1,1], resize to the current resolution: one fused pass on the GPU
            x_real = grower.transform_batch(x_real.to(device, non_blocking=True))
            alpha = grower.alpha
            d = stepper.d_step(x_real, alpha)
            g = stepper.g_step(batch_size, alpha, device) if iter_idx % 5 == 0 else None
            metrics.push(d, g)

            if iter_idx % metric_every == 0:
                metrics.flush()  # one device->host copy for the whole window
                if rank == 0 and hasattr(bar, "set_description"):
                    h = metrics.hist
                    bar.set_description(
                        f"Epoch {e:02} [{saver.curr_save:03}: {saver.save_counter:03}], "
                        f"disc_l = {mean(h['disc_loss']):.4f}, gen_l = {mean(h['gen_loss']):.2f}, "
                        f"grad_p = {mean(h['grad_pen']):.4f}, e_tp = {mean(h['e_tp']):.2f}, "
                        f"e_tn = {mean(h['e_tn']):.2f}, e_gen = {mean(h['e_gen']):.2f}, alpha = {alpha:.3f}")
            if iter_idx % 200 == 0 and mlflow is not None and rank == 0:
                metrics.flush()
                mlflow.log_metrics(step=gen.curr_layer, metrics={
                    "disc_loss": metrics.last["disc_loss"], "gen_loss": metrics.last["gen_loss"],
                    "batch_tp_error": metrics.last["e_tp"], "batch_tn_error": metrics.last["e_tn"]})

            iter_idx += 1
            pos += 1
            if grower.grow(batch_size * world) and gen.growing:
                stepper.finish()
                grow_networks()
                if rank == 0 and hasattr(bar, "write"):
                    bar.write(f"\nNext layer, {gen.curr_layer} / {gen.down_sample}, curr_save = {saver.curr_save}")

            # checkpoint AFTER the growth bookkeeping (the reference saves in front of it, train.py:248-272, but it never
            # resumes): the saved networks, optimizers, Grower counters and iteration index then describe one consistent
            # point -- "iteration iter_idx is the next one to run"
            if saver.due():
                stepper.finish()
                rng_states = noise_states()
                if rank == 0:
                    saver.request_save(gen, disc, optim_gen, optim_disc, alpha, train_state=lambda: {
                        "grower": grower.state_dict(), "level": gen.curr_layer, "iter_idx": iter_idx, "epoch": e,
                        "epoch_pos": pos, "base_seed": base_seed, "noise_rng": rng_states, "world": world,
                        "saver": saver.state_dict_after_save()})
                else:
                    saver.tick()
            else:
                saver.tick()
            if progress_hook is not None:
                progress_hook(iter_idx)
            if max_iters and iter_idx >= max_iters:
                break
        else:
            continue
        break
    stepper.finish()
    metrics.flush()
    if mlflow is not None and rank == 0:
        mlflow.end_run()
        mlflow.end_run()
