"""`train(run_name, input_dataset_path, output_dir)` with the reference's signature, hyper-parameters, growth schedule, D:G
cadence and checkpoint cadence (/root/reference/music_gan/train.py:18-278) on the MI355X stepper.

Differences, all outside the arithmetic: mlflow is optional; loss read-backs happen every `metric_every` iterations instead of
4-6 `.item()` syncs per iteration; launched under torchrun (WORLD_SIZE > 1) it runs data-parallel, one process per GPU, with
the dataset sharded by a DistributedSampler and the growth schedule advanced by the GLOBAL sample count so all ranks grow on
the same iteration."""
from __future__ import annotations

import os
from os import mkdir
from os.path import exists, isdir
from statistics import mean

import torch as th
from torch.utils.data import DataLoader

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


def train(run_name: str, input_dataset_path: str, output_dir: str, *, nb_epoch: int = 1000, batch_size: int = 6,
          num_workers: int = 6, metric_every: int = 20, max_iters: int = 0, save_every: int = 1000,
          resume_from: str = None) -> None:
    """Reference signature plus keyword-only extensions (all defaulting to the reference's literals).  `resume_from`: a
    directory written by a previous run; the newest `train_state_k.pt` / `gen_k.pt` / `disc_k.pt` / `optim_*_k.pt` set is
    loaded (growth level, Grower counters, weights, Adam state) -- the reference can only save (train.py never loads)."""
    assert isdir(input_dataset_path), f"\"{input_dataset_path}\" doesn't exist or is not a directory"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    th.cuda.set_device(local_rank)
    device = th.device("cuda", local_rank)
    if world > 1 and not th.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        th.distributed.init_process_group(backend="nccl", device_id=device)
    if mlflow is not None and rank == 0:
        mlflow.set_experiment("music_gan")
        mlflow.start_run(run_name=run_name)

    sample_rate = audio.SAMPLE_RATE
    rand_channels, height, width = 32, 2, 2
    disc_lr = gen_lr = 1e-3
    betas = (0.0, 0.9)

    if not exists(output_dir):
        if rank == 0:
            mkdir(output_dir)
    elif not isdir(output_dir):
        raise NotADirectoryError(f"\"{output_dir}\" is not a directory !")

    th.manual_seed(0 if world > 1 else th.initial_seed())
    gen = networks.Generator(rand_channels, end_layer=0).to(device)
    disc = networks.Discriminator(start_layer=7).to(device)
    from .dist import broadcast_parameters
    broadcast_parameters([gen, disc])
    optim_gen = FusedAdam(gen.parameters(), lr=gen_lr, betas=betas)
    optim_disc = FusedAdam(disc.parameters(), lr=disc_lr, betas=betas)
    start_iter = 0
    resume_state = None
    if resume_from is not None:
        import glob
        import re
        states = glob.glob(os.path.join(resume_from, "train_state_*.pt"))
        assert states, f"no train_state_*.pt in \"{resume_from}\""
        k = max(int(re.search(r"train_state_(\d+)\.pt$", p).group(1)) for p in states)
        resume_state = th.load(os.path.join(resume_from, f"train_state_{k}.pt"))
        for _ in range(resume_state["level"]):  # replay the growth so parameter groups line up with the saved optimizers
            gen.next_layer()
            disc.next_layer()
            optim_gen.add_param_group({"params": gen.end_block_params(), "lr": gen_lr, "betas": betas})
            optim_disc.add_param_group({"params": disc.start_block_parameters(), "lr": disc_lr, "betas": betas})
        gen.load_state_dict(th.load(os.path.join(resume_from, f"gen_{k}.pt"), map_location=device))
        disc.load_state_dict(th.load(os.path.join(resume_from, f"disc_{k}.pt"), map_location=device))
        optim_gen.load_state_dict(th.load(os.path.join(resume_from, f"optim_gen_{k}.pt"), map_location=device))
        optim_disc.load_state_dict(th.load(os.path.join(resume_from, f"optim_disc_{k}.pt"), map_location=device))
        for opt in (optim_gen, optim_disc):  # torch keeps `step` on the host, the moments on the device
            for st in opt.state.values():
                st["step"] = st["step"].to("cpu")
        start_iter = int(resume_state["iter_idx"])
    stepper = ProGANStepper(gen, disc, optim_gen, optim_disc, rand_channels, height, width)

    audio_dataset = audio.AudioDataset(input_dataset_path)
    sampler = None
    if world > 1:
        sampler = th.utils.data.distributed.DistributedSampler(audio_dataset, shuffle=True, drop_last=True)
    data_loader = DataLoader(audio_dataset, batch_size=batch_size, shuffle=sampler is None, sampler=sampler,
                             num_workers=num_workers, drop_last=True, pin_memory=True)

    if mlflow is not None and rank == 0:
        mlflow.log_params({"input_dataset": input_dataset_path, "nb_sample": len(audio_dataset),
                           "output_dir": output_dir, "rand_channels": rand_channels, "nb_epoch": nb_epoch,
                           "batch_size": batch_size, "disc_lr": disc_lr, "gen_lr": gen_lr, "betas": betas,
                           "sample_rate": sample_rate, "width": width, "height": height, "world_size": world})

    grower = Grower(n_grow=7, fadein_lengths=[1, 25000, 37500, 50000, 62500, 75000, 87500, 100000],
                    train_lengths=[50000, 100000, 150000, 200000, 250000, 300000, 350000])
    saver = Saver(output_dir, save_every=save_every, rand_channels=rand_channels, rand_height=height, rand_width=width)
    if resume_state is not None:
        grower.load_state_dict(resume_state["grower"])

    window = 20
    hist = {k: [0.] * window for k in ("disc_loss", "grad_pen", "gen_loss", "e_tp", "e_tn", "e_gen")}
    pending = []  # device scalars waiting for the next metric read-back
    iter_idx = start_iter
    last_gen = None
    for e in range(nb_epoch):
        if sampler is not None:
            sampler.set_epoch(e)
        bar = tqdm(data_loader) if rank == 0 else data_loader
        for x_real in bar:
            # float64 -> float32, per-channel min-max to [-1,1], resize to the current resolution: one fused pass on the GPU
            x_real = grower.transform_batch(x_real.to(device, non_blocking=True))
            alpha = grower.alpha
            d = stepper.d_step(x_real, alpha)
            g = None
            if iter_idx % 5 == 0:
                g = stepper.g_step(batch_size, alpha, device)
                last_gen = g
            pending.append((d, g))

            if iter_idx % metric_every == 0:
                for dm, gm in pending:  # one sync for the whole window
                    for key, src in (("disc_loss", "disc_loss"), ("grad_pen", "grad_pen"), ("e_tp", "out_real_mean"),
                                     ("e_tn", "out_fake_mean")):
                        hist[key] = hist[key][1:] + [float(dm[src])]
                    if gm is not None:
                        hist["gen_loss"] = hist["gen_loss"][1:] + [float(gm["gen_loss"])]
                        hist["e_gen"] = hist["e_gen"][1:] + [float(gm["out_fake_mean"])]
                pending.clear()
                if rank == 0 and hasattr(bar, "set_description"):
                    bar.set_description(
                        f"Epoch {e:02} [{saver.curr_save:03}: {saver.save_counter:03}], "
                        f"disc_l = {mean(hist['disc_loss']):.4f}, gen_l = {mean(hist['gen_loss']):.2f}, "
                        f"grad_p = {mean(hist['grad_pen']):.4f}, e_tp = {mean(hist['e_tp']):.2f}, "
                        f"e_tn = {mean(hist['e_tn']):.2f}, e_gen = {mean(hist['e_gen']):.2f}, alpha = {alpha:.3f}")
            if iter_idx % 200 == 0 and mlflow is not None and rank == 0 and last_gen is not None:
                mlflow.log_metrics(step=gen.curr_layer, metrics={
                    "disc_loss": float(d["disc_loss"]), "gen_loss": float(last_gen["gen_loss"]),
                    "batch_tp_error": float(d["out_real_mean"]), "batch_tn_error": float(d["out_fake_mean"])})

            iter_idx += 1
            if rank == 0:
                if (saver.save_counter + 1) % save_every == 0:
                    stepper.finish()
                saver.request_save(gen, disc, optim_gen, optim_disc, alpha, train_state=lambda: {
                    "grower": grower.state_dict(), "level": gen.curr_layer, "iter_idx": iter_idx})

            if grower.grow(batch_size * world) and gen.growing:
                stepper.finish()
                th.manual_seed(1000 + gen.curr_layer)  # identical new head/stem on every rank
                gen.next_layer()
                disc.next_layer()
                optim_gen.add_param_group({"params": gen.end_block_params(), "lr": gen_lr, "betas": betas})
                optim_disc.add_param_group({"params": disc.start_block_parameters(), "lr": disc_lr, "betas": betas})
                if rank == 0 and hasattr(bar, "write"):
                    bar.write(f"\nNext layer, {gen.curr_layer} / {gen.down_sample}, curr_save = {saver.curr_save}")
            if max_iters and iter_idx >= max_iters:
                stepper.finish()
                return
    stepper.finish()
