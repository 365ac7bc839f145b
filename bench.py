#!/usr/bin/env python3
"""Headline benchmark: spectrogram-images/sec of one ProGAN G+D step (one discriminator update with gradient penalty + one
generator update, Adam on both -- /root/reference/music_gan/train.py:143-175,191-214) on synthetic 2x128x128 STFT tensors,
batch 64 per GPU, level 5 with the fade-in branch live (alpha = 0.5).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (whole-step algorithmic FLOPs against the
exact-fp32 MFMA peak, plus the dominant kernel timed live with HIP events) and, at N=1, `cpu_baseline` (the CPU oracle -- a
plain-PyTorch port of the reference step as executed -- on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, exact fp32
LEVEL_SIDE = {3: 32, 4: 64, 5: 128, 6: 256, 7: 512}


def build_nets(level: int, rand_channels: int, device):
    from musicgan_amd.networks import Discriminator, Generator
    torch.manual_seed(0)
    gen, disc = Generator(rand_channels, end_layer=0), Discriminator(start_layer=7)
    for _ in range(level):
        gen.next_layer()
        disc.next_layer()
    return gen.to(device), disc.to(device)


def flops_per_image(level: int, rand_channels: int) -> float:
    """Algorithmic minimum 4*Gf + 12*Df; conv/linear MACs x 2 only.  Pure arithmetic on the layer table."""
    tail = [128, 112, 96, 80, 64, 48, 32, 16]
    ins = [rand_channels] + tail[:-1]
    dch = [(16, 32), (32, 48), (48, 64), (64, 80), (80, 96), (96, 112), (112, 128), (128, 144), (144, 160)]
    g, s = 0.0, 2
    for i in range(level + 1):
        g += 18 * ins[i] * ins[i] * s * s
        s *= 2
        g += 18 * ins[i] * tail[i] * s * s
    g += 4 * tail[level] * s * s + (4 * tail[level - 1] * (s // 2) ** 2 if level > 0 else 0)
    dl = 7 - level
    d = 4 * dch[dl][0] * s * s + (4 * dch[dl][1] * (s // 2) ** 2 if level > 0 else 0)
    t = s
    for i in range(dl, 9):
        ci, co = dch[i]
        d += 18 * ci * co * t * t
        t //= 2
        d += 18 * co * co * t * t
    d += 320
    return 4 * g + 12 * d


def executed_flops_per_image(level: int, rand_channels: int, batch: int) -> float:
    """FLOPs the MFMA pipe actually executes per image and D+G step: the same layer walk as `flops_per_image`, each 3x3
    convolution pass weighted 1/2.25 where the engine runs it in Winograd F(2x2,3x3) / F(3x3,2x2) or sub-pixel form, 1/4 where an
    up-sampling layer runs in 9-component Winograd form (csrc/wino_ups.hip; weight gradient: (1/2.25)(9/16)) (the kernel
    choice predicates of musicgan_amd.ops, evaluated for the batch each pass really sees: 3N in the fused critic step, N
    elsewhere) and 1 where it takes the direct implicit GEMM.  Static arithmetic; padding of odd channel counts not counted."""
    from musicgan_amd import ops
    from musicgan_amd.networks.engine import gen_conv_exec_factor
    tail = [128, 112, 96, 80, 64, 48, 32, 16]
    ins = [rand_channels] + tail[:-1]
    dch = [(16, 32), (32, 48), (48, 64), (64, 80), (80, 96), (96, 112), (112, 128), (128, 144), (144, 160)]
    n = batch
    W = 1.0 / 2.25

    def conv(nb, cin, cout, h, w, pixnorm=False):  # forward-like pass (forward, data gradient, tangent)
        return 18.0 * cin * cout * h * w * nb * (W if ops.wino3x3_supported(nb, cout, h, w, pixnorm=pixnorm, cin=cin) else 1.0)

    def wgrad(nb, cin, cout, h, w, ups=False):
        if not ops.wino_wgrad_supported(nb, cin, cout, h, w, ups=ups):
            return 18.0 * cin * cout * h * w * nb
        nine = ops.wino_wgrad_form(nb, cin, cout, h, w, ups=ups) == 2  # up-sampled input, row-staged kernel: 9 of the 16 products
        return 18.0 * cin * cout * h * w * nb * W * (9.0 / 16.0 if nine else 1.0)

    def gen_pass(backward: bool) -> float:
        f, s = 0.0, 2
        for i in range(level + 1):
            ci, co = ins[i], tail[i]
            # the two PixelNorm convs of a block: the form the engine really runs (engine.gen_conv_form, shared with PackCache)
            f += 18.0 * ci * ci * s * s * n * gen_conv_exec_factor(n, ci, ci, s, s, False)
            f += 18.0 * ci * co * 4 * s * s * n * gen_conv_exec_factor(n, ci, co, s, s, True)
            if backward:
                f += wgrad(n, ci, co, 2 * s, 2 * s, ups=True) + wgrad(n, ci, ci, s, s)
                f += 18.0 * ci * co * 4 * s * s * n * (0.25 if ops.winoups3x3_supported(n, ci, co, s, s, dgrad=True) else
                                                       (W if ops.upconv3x3_dgrad_supported(s, s, n * co * 4 * s * s, n) else 1.0))
                if i > 0:
                    f += conv(n, ci, ci, s, s)
            s *= 2
        return f + 4.0 * n * (tail[level] * s * s + (tail[level - 1] * (s // 2) ** 2 if level > 0 else 0)) * (3 if backward else 1)

    def disc_pass(nb, passes_fwd_like: int, with_wgrad: bool) -> float:
        f, t = 0.0, LEVEL_SIDE[level]
        for i in range(7 - level, 9):
            ci, co = dch[i]
            f += passes_fwd_like * (conv(nb, ci, co, t, t) + conv(nb, co, co, t // 2, t // 2))
            if with_wgrad:
                f += wgrad(nb, ci, co, t, t) + wgrad(nb, co, co, t // 2, t // 2)
            t //= 2
        side = LEVEL_SIDE[level]
        small = 4.0 * nb * (dch[7 - level][0] * side * side + (dch[7 - level][1] * (side // 2) ** 2 if level > 0 else 0)) + 320.0 * nb
        return f + small * (passes_fwd_like + (1 if with_wgrad else 0))

    d_step = gen_pass(False) + disc_pass(3 * n, 2, True) + disc_pass(n, 1, False)  # fwd + dgrad over 3N, wgrad over 3N, tangent over N
    g_step = gen_pass(True) + disc_pass(n, 2, False)
    return (d_step + g_step) / n


def dominant_kernel_probe(device, batch: int, iters: int = 100):
    """Time the dominant kernel of the step -- the first discriminator conv 48->64 @128x128 + LeakyReLU + fused AvgPool2d over the
    fused critic step's batch [real|fake|interpolated] = 3*batch images, Winograd F(2x2,3x3) kernel wino3x3_strip<2,8,ACT_POOL_MOUT>
    (csrc/wino_strip.hip; rounds 1-4: wino3x3_mfma<2,2,4>) (173.9 algorithmic GFLOP at batch 64) -- with HIP events on the stream it is launched on.  `tflops` is ALGORITHMIC (18*Cin*Cout FLOP
    per output pixel, the direct-convolution count the roofline is defined on); the kernel executes 2.25x fewer multiplies."""
    from musicgan_amd import ops
    g = torch.Generator(device=device).manual_seed(1)
    n = 3 * batch
    x = torch.randn(n, 48, 128, 128, device=device, generator=g)
    w = torch.randn(64, 48, 3, 3, device=device, generator=g) * 0.04
    b = torch.randn(64, device=device, generator=g)
    up = ops.pack_wino3x3(w, dgrad=False)
    q = torch.empty(n, 64, 64, 64, device=device)
    # as engine.disc_forward calls it: pooled result + one sign byte per 2x2 tile (the full-resolution activation is not written)
    fn = lambda: ops.conv3x3(x, None, b, 64, lrelu=True, pool_out=q, wino=up, mask_out=True)
    for _ in range(30):  # (the first launches after host-side work run 10-25 % slow: 989 us max against 783 avg of 300 in the profile)
        fn()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2.0 * 9 * 48 * 64 * 128 * 128 * n
    out = {"name": "wino3x3_strip<2,8,ACT_POOL_MOUT> Winograd F(2x2,3x3) conv 48->64@128x128 + lrelu + avgpool + tile mask, 3x batch", "ms": ms,
           "flop": flop, "tflops": flop / ms / 1e9, "executed_tflops": flop / 2.25 / ms / 1e9,
           "algorithmic_bytes": n * (4.0 * 48 * 128 * 128 + 4.0 * 64 * 64 * 64 + 1.0 * 64 * 64 * 64)}
    # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), measured by tools/measure_traffic.sh
    tpath = os.path.join(ROOT, "profiles", "traffic_dominant_kernel.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            out["hbm_traffic"] = json.load(f)
    return out


def time_step(level: int, batch: int, rand_channels: int, device, steps: int, warmup: int, seed: int = 1234, alpha: float = 0.5):
    """ms per D+G step of a fresh single-GPU stepper at (level, batch), HIP events on the launch stream, nothing skipped."""
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper
    gen, disc = build_nets(level, rand_channels, device)
    og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    side = LEVEL_SIDE[level]
    rng = torch.Generator(device=device).manual_seed(seed)
    stepper = ProGANStepper(gen, disc, og, od, rand_channels, noise=rng)
    x_real = torch.rand(batch, 2, side, side, device=device, generator=rng) * 2 - 1

    def one():
        stepper.d_step(x_real, alpha)
        stepper.g_step(batch, alpha, device)

    for _ in range(warmup):
        one()
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        one()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / steps


def step_roofline(fpi: float, xfpi: float, ips_per_gpu: float) -> dict:
    """`frac` = `achieved` / `peak` with achieved = the FLOPs the MFMA pipe really EXECUTES per second (Winograd / sub-pixel passes
    count 1/2.25): a fraction of the machine, never above 1.  `algorithmic_frac` counts the direct-convolution FLOPs the step
    replaces (SURVEY 8(d)'s 4*Gf + 12*Df definition) -- an efficiency figure that may exceed `frac` by up to 2.25x."""
    return {"bound": "mfma", "achieved": xfpi * ips_per_gpu / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": xfpi * ips_per_gpu / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "algorithmic_achieved": fpi * ips_per_gpu / 1e12,
            "algorithmic_frac": fpi * ips_per_gpu / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "basis": f"{xfpi / 1e9:.2f} executed / {fpi / 1e9:.2f} algorithmic GFLOP per image x images/s per GPU"}


def step_traffic(level: int, batch: int):
    """Whole-step HBM bytes (rocprofv3 PMC over D+G steps: FETCH_SIZE x 2 + WRITE_SIZE summed over every dispatch, collected by
    tools/measure_step_traffic.sh and committed under profiles/) beside SURVEY 8(d)'s ~0.7 GB / image estimate."""
    tpath = os.path.join(ROOT, "profiles", f"traffic_step_l{level}.json")
    if not os.path.exists(tpath):
        return None
    with open(tpath) as f:
        rec = json.load(f)
    return rec if rec.get("batch") == batch else None


def level_record(device, rand_channels: int, level: int, batch: int, steps: int, warmup: int, what: str, alpha: float = 0.5):
    """One more (level, batch) with the headline's step and accounting: BASELINE.json configs[0..1] and the levels a real run of
    the reference lives at -- it trains at batch 6 (train.py:43) and spends 65 % of its scheduled FLOPs at level 6 and everything
    after 1.4 M samples at level 7 (train.py:101-109)."""
    ms = time_step(level, batch, rand_channels, device, steps=steps, warmup=warmup, alpha=alpha)
    ips = batch / ms * 1e3
    side = LEVEL_SIDE[level]
    fpi, xfpi = flops_per_image(level, rand_channels), executed_flops_per_image(level, rand_channels, batch)
    return {"workload": f"ProGAN level {level} WGAN-GP D+G step, 2x{side}x{side}, batch {batch}, alpha {alpha:g} ({what})",
            "value": ips, "unit": "images/s", "ms_per_step": ms, "steps": steps, "warmup": warmup,
            "roofline": step_roofline(fpi, xfpi, ips)}


def stft_record(device, cpu: bool):
    """BASELINE.json configs[4] on one GPU: one 10-minute 44.1 kHz file (mono U(-0.5,0.5), seed 7; 103 360 frames) through
    mg_stft_1024 (audio/functions.py:38-62) and through STFT + codec (wav_to_stft + stft_to_phase_magn, :38-94), HIP events on the
    launch stream, input resident in HBM.  Roofline: 5 120 algorithmic bytes per frame (1 024 B of new samples in, 4 096 B of
    complex bins out) against 8 TB/s."""
    from musicgan_amd import audio, ops
    L = 44100 * 600
    g = torch.Generator(device=device).manual_seed(7)
    wav = torch.rand(L, device=device, generator=g) - 0.5
    T = 1 + L // 256

    def timeit(fn, iters, warm=3):
        for _ in range(warm):
            fn()
        stream = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters

    # steady state as for the training step (>= 10 warm-up, >= 50 timed).  This HBM-bound kernel needs ~25 ms of load
    # (~200 launches) after host-side work before the chip's memory-side clocks have ramped: batches of 50 launches after 2 s of
    # idle measure 0.44, 0.48, 0.52, 0.53, 0.54, 0.55, 0.55 ... of the HBM roofline (tools/bench_stft_ramp.py)
    # The ramp is not always the same length (what ran before matters: 250 launches were enough after the level-4 record on
    # most boxes and left the record at 0.156 ms instead of 0.118 on one), so the warm-up is adaptive: batches of 50 launches until
    # two successive batches agree within 1.5 % (at least 5, at most 40 batches), then 200 timed launches.
    prev, settled = None, 0
    for b in range(40):
        cur = timeit(lambda: ops.stft_1024(wav), 50, warm=0)
        settled = settled + 1 if (prev is not None and abs(cur - prev) <= 0.015 * prev) else 0
        prev = cur
        if b >= 4 and settled >= 2:
            break
    ms_stft = timeit(lambda: ops.stft_1024(wav), 200, warm=0)
    ms_both = timeit(lambda: audio.stft_to_phase_magn(ops.stft_1024(wav)), 50, warm=10)
    c_keep = ops.stft_1024(wav)
    mp = torch.stack(audio.stft_to_phase_magn(ops.stft_1024(wav)), dim=1)[:8].contiguous()  # 8 samples = 4 096 frames
    fps = T / (ms_stft * 1e-3)
    rec = {"workload": "10 min mono 44.1 kHz synthetic file, n_fft 1024 hop 256: 103 360 frames -> 201 samples (2x512x512)",
           "metric": "STFT frames/s (mg_stft_1024)", "value": fps, "unit": "frames/s", "ms_per_file": ms_stft,
           "stft_plus_codec": {"ms_per_file": ms_both, "frames_per_s": T / (ms_both * 1e-3),
                               "samples_per_s": ((T - 1) // 512) / (ms_both * 1e-3)},
           "roofline": {"bound": "hbm", "achieved": 5120.0 * fps / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": 5120.0 * fps / 8e12, "traffic": stft_traffic(),
                        "algorithmic_bytes": 5120.0 * T,
                        "basis": "5 120 algorithmic bytes per frame x frames per launch / HIP-event launch time; traffic = "
                                 "HBM bytes per launch, NOT measured in this run: read from profiles/traffic_stft_kernel.json "
                                 "(rocprofv3 PMC passes, tools/measure_traffic.sh stft, re-measured in round 6)"}}
    ms_codec = timeit(lambda: audio.stft_to_phase_magn(c_keep), 50, warm=10)
    rec["codec"] = {"workload": "stft_to_phase_magn (functions.py:65-94) of that file's 512 x 103 360 bins -> 201 images",
                    "ms_per_file": ms_codec,
                    "roofline": {"bound": "hbm", "achieved": 16.0 * 512 * T / ms_codec / 1e6, "peak": 8000.0, "unit": "GB/s",
                                 "frac": 16.0 * 512 * T / (ms_codec * 1e-3) / 8e12,
                                 "basis": "16 algorithmic bytes per bin (8 B complex in, 2 x 4 B out; SURVEY 8a S2); the global "
                                          "min/max forces a second pass, so the kernels move 32 B per bin"}}
    ms_inv = timeit(lambda: audio.functions.magn_phase_to_waveform(mp), 10)
    rec["inverse"] = {"workload": "8 samples (4 096 frames) -> waveform, mg_codec_inv (functions.py:97-139)",
                      "ms": ms_inv, "frames_per_s": 4096 / (ms_inv * 1e-3)}
    if cpu:
        rec["cpu_baseline"] = stft_cpu_baseline(wav)
    return rec


def stft_traffic():
    """HBM bytes one stft1024_kernel launch moves (FETCH_SIZE x 2 + WRITE_SIZE from separate rocprofv3 --pmc passes, collected
    by `bash tools/measure_traffic.sh stft` and committed under profiles/)."""
    tpath = os.path.join(ROOT, "profiles", "traffic_stft_kernel.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            return json.load(f).get("bytes_per_launch")
    return None


def create_dataset_and_train_records(device, rand_channels: int):
    """BASELINE.json configs[4] end to end, and the real training loop on what it wrote.

    `create_dataset_e2e`: musicgan_amd.create_dataset (reference create_dataset.py:34-64) on two synthetic 10-minute mono
    44.1 kHz wav files in a scratch directory: files/s, samples/s and where the wall time goes -- the GPU part is ~0.5 ms per file,
    the loop is bound by the 843 MB of float64 `.pt` files per file that the reference's format demands.
    `train_loop`: musicgan_amd.train.train (reference train.py:131-272) on that dataset -- packed loader, device input transform,
    metric window, the reference's 5 critic : 1 generator cadence -- with the growth schedule overridden so that it grows to level 5
    in its first five iterations and stays there, batch 64; 200 iterations timed between two synchronised marks (all graphs
    captured by the first mark).  Comparable with `secondary` (the same cadence on resident synthetic tensors)."""
    import glob
    import shutil
    import tempfile
    from musicgan_amd import create_dataset
    from musicgan_amd.audio import wavio
    from musicgan_amd.train import train
    tmp = tempfile.mkdtemp(prefix="mg_bench_")
    out = {}
    try:
        wav_dir, data = os.path.join(tmp, "wav"), os.path.join(tmp, "data")
        os.mkdir(wav_dir)
        g = torch.Generator().manual_seed(7)
        nfiles, nlong = 2, 8  # (2.8 GB of scratch for the two-file data set the training loop then reads; 7.6 GB, deleted at once, for eight)
        for i in range(nlong):
            wavio.save(os.path.join(wav_dir, f"track_{i}.wav"), torch.rand(1, 44100 * 600, generator=g) - 0.5, 44100)

        def record(pattern, folder, n):
            stats = {}
            t0 = time.perf_counter()
            create_dataset(os.path.join(wav_dir, pattern), folder, stats=stats)
            wall = time.perf_counter() - t0
            return stats, {
                "workload": f"create_dataset on {n} synthetic 10-minute mono 44.1 kHz float32 wav files -> "
                            f"{stats['samples']} float64 (2,512,512) .pt files + float32 side-car, scratch dir {tmp}",
                "files_per_s": n / wall, "samples_per_s": stats["samples"] / wall, "wall_s": wall,
                "pt_MB_per_s": stats["pt_bytes"] / wall / 1e6,
                # the call's fixed cost (page-locking the 256 MiB chunk ring + the loader's staging buffers, starting the threads):
                "files_per_s_after_setup": n / (wall - stats["setup_s"]),
                "split_s": {"setup_pinned_ring_and_threads": stats["setup_s"],
                            "waiting_for_the_loader_thread_and_stft_launch": stats["load_stft_s"],
                            "loader_threads": stats["loader_threads"],
                            "loader_busy_thread_s_read_pin_upload": stats["loader_thread_busy_s"],
                            "codec_d2h_submit": stats["codec_copy_submit_s"], "waiting_for_the_writers_at_the_end": stats["drain_s"],
                            "of_which_waiting_for_a_free_pinned_chunk": stats["ring_wait_s"],
                            "writer_threads": stats["writer_threads"], "writer_busy_thread_s": stats["writer_busy_s"]}}
        stats, out["create_dataset_e2e"] = record("track_[01].wav", data, nfiles)
        try:
            st8, out["create_dataset_e2e_8_files"] = record("track_*.wav", os.path.join(tmp, "data8"), nlong)
            # the bound of this box, measured in the same run and the same scratch directory, and how much of it the loop reaches
            probe = host_io_probe(device, tmp, st8["writer_threads"])
            rec8 = out["create_dataset_e2e_8_files"]
            rec8["roofline"] = {"bound": "host io", "achieved": rec8["files_per_s_after_setup"], "peak": probe["bound_files_per_s"],
                                "unit": "files/s", "frac": rec8["files_per_s_after_setup"] / probe["bound_files_per_s"], "probe": probe}
        except OSError as e:  # (scratch space: the record is optional)
            out["create_dataset_e2e_8_files"] = {"skipped": str(e)}
        shutil.rmtree(os.path.join(tmp, "data8"), ignore_errors=True)
        shutil.rmtree(wav_dir)
        marks = {}
        first, last = 30, 230

        def hook(it):
            if it in (first, last):
                torch.cuda.synchronize()
                marks[it] = time.perf_counter()

        big = 10 ** 9
        devnull = open(os.devnull, "w")
        stderr, stdout, sys.stderr, sys.stdout = sys.stderr, sys.stdout, devnull, devnull  # tqdm's bar and its "Next layer" lines
        try:
            train("bench", data, os.path.join(tmp, "out"), nb_epoch=1000, batch_size=64, max_iters=last, save_every=big,
                  rand_channels=rand_channels, fadein_lengths=[1, 1, 1, 1, 1, 10 ** 7, big, big],
                  train_lengths=[1, 1, 1, 1, 1, big, big], progress_hook=hook)
        finally:
            sys.stderr, sys.stdout = stderr, stdout
            devnull.close()
        dt = marks[last] - marks[first]
        out["train_loop"] = {
            "workload": "musicgan_amd.train.train parked at level 5 (2x128x128), batch 64: packed loader + device input transform + "
                        "5 critic : 1 generator updates + metric window, 200 iterations between synchronised marks",
            "iterations": last - first, "ms_per_iteration": 1e3 * dt / (last - first),
            "images_per_s": 64 * (last - first) / dt, "dataset_samples": stats["samples"]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def host_io_probe(device, scratch_dir: str, threads: int, samples_per_thread: int = 48) -> dict:
    """What bounds create_dataset end to end on THIS box (BASELINE configs[4]; reference create_dataset.py:34-64 leaves 201 float64
    (2,512,512) .pt files = 843 MB per 10-minute file behind, ours additionally the 421 MB float32 side-car): (a) the pinned
    device-to-host rate in the loop's 64 MiB chunks, one and two copy streams; (b) the writer path of one sample -- float32 ->
    float64, `writev` of the 4 MiB .pt file, `pwrite` of the 2 MiB side-car row -- on one thread and on `threads` threads,
    in the scratch directory the record uses.  bound per file = max(421 MB / D2H rate, 201 samples / writer-path rate)."""
    import threading
    chunk_bytes = 32 * 2 * 512 * 512 * 4
    src = torch.empty(chunk_bytes // 4, dtype=torch.float32, device=device).normal_()
    pins = [torch.empty(chunk_bytes // 4, dtype=torch.float32).pin_memory() for _ in range(4)]
    out = {"chunk_MiB": chunk_bytes >> 20}
    for nstreams in (1, 2):
        streams = [torch.cuda.Stream(device=device) for _ in range(nstreams)]
        reps = 16
        for k in range(2):  # (first round: warm-up)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                with torch.cuda.stream(streams[i % nstreams]):
                    pins[i % 4].copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[f"d2h_GB_per_s_{nstreams}_stream{'s' if nstreams > 1 else ''}"] = reps * chunk_bytes / dt / 1e9
    # the writer threads' own native call (mg_pt_write_samples: side-car row + widening + writev, 8 samples per call, no interpreter
    # lock) on pinned chunks the copy engine has just filled -- cold for the CPU; every thread its own side-car file, as the shards
    import ctypes
    from musicgan_amd import _lib
    lib = _lib.load()
    prefix, suffix = b"p" * 1024, b"s" * 600  # (the container's bytes around the payload: about the sizes of fast_pt.PtTemplate's)
    row_floats = 2 * 512 * 512
    batches = max(1, samples_per_thread // 8)

    def writer(tid, busy):
        side = os.open(os.path.join(scratch_dir, f"probe_side_{tid}.bin"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        t0 = time.perf_counter()
        for bi in range(batches):
            r0 = ((tid * 3 + bi) % 4) * 8
            paths = b"\0".join(os.fsencode(os.path.join(scratch_dir, f"probe_{tid}_{bi * 8 + k}.pt")) for k in range(8)) + b"\0"
            rc = lib.mg_pt_write_samples(ctypes.c_void_p(pins[(tid + bi) % 4].data_ptr() + r0 * row_floats * 4), 8, row_floats, paths,
                                         prefix, len(prefix), suffix * 8, len(suffix), side, bi * 8 * row_floats * 4)
            assert rc == 0
        busy[tid] = time.perf_counter() - t0
        os.close(side)

    # (the independent loops run BEFORE the product's own path: each leaves gigabytes of dirty pages behind, and whoever writes second
    # is throttled by the page cache -- measured the other way round, the "bound" came out 1.4x slower than the product)
    # The bound itself must not be the product's own call: the two things a sample costs the host whatever the code around them --
    # (i) plain `write` of the sample's 4 MiB into a new file + `pwrite` of its 2 MiB side-car row into a per-thread file, from a zero buffer (page cache, inode
    # and directory work of this file system), (ii) a float32 -> float64 widening of 2 x 512 x 512 values -- each on `threads`
    # threads at once, as native loops (mg_host_io_probe: through ctypes without the interpreter lock; the same calls made from
    # Python threads measured the lock hand-over, 2x slower than the product's writer path).
    import numpy as np
    src32 = [np.random.default_rng(t).random(row_floats, dtype=np.float32) for t in range(threads)]
    n_ind = batches * 8  # (as many samples per thread as the product path below)
    wsec, dsec = [0.0] * threads, [0.0] * threads

    def prober(tid):
        ws, ds = ctypes.c_double(0.0), ctypes.c_double(0.0)
        rc = lib.mg_host_io_probe(os.fsencode(scratch_dir), tid, n_ind, row_floats * 8 + len(prefix) + len(suffix), row_floats * 4, ctypes.c_void_p(src32[tid].ctypes.data), row_floats,
                                  ctypes.byref(ws), ctypes.byref(ds))
        assert rc == 0, lib.mg_last_error()
        wsec[tid], dsec[tid] = ws.value, ds.value

    ths = [threading.Thread(target=prober, args=(t,)) for t in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    # (every thread runs its n_ind writes, then its n_ind widenings; seconds per sample and thread with all threads busy)
    indep = {"raw_write": max(wsec) / n_ind, "widen": max(dsec) / n_ind}
    out[f"raw_write_ms_per_sample_{threads}_threads"] = 1e3 * indep["raw_write"]
    out[f"widen_ms_per_sample_{threads}_threads"] = 1e3 * indep["widen"]
    for t in range(threads):
        for i in range(n_ind):
            os.remove(os.path.join(scratch_dir, f"probe_raw_{t}_{i}.bin"))
        os.remove(os.path.join(scratch_dir, f"probe_rawside_{t}.bin"))
    for nthr in (1, threads):
        busy = [0.0] * nthr
        ths = [threading.Thread(target=writer, args=(t, busy)) for t in range(nthr)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        wall = time.perf_counter() - t0
        out[f"writer_path_samples_per_s_{nthr}_thread{'s' if nthr > 1 else ''}"] = nthr * batches * 8 / wall
        out[f"writer_path_ms_per_sample_{nthr}_thread{'s' if nthr > 1 else ''}"] = 1e3 * sum(busy) / (nthr * batches * 8)
        for t in range(nthr):
            for i in range(batches * 8):
                os.remove(os.path.join(scratch_dir, f"probe_{t}_{i}.pt"))
            os.remove(os.path.join(scratch_dir, f"probe_side_{t}.bin"))
    d2h = max(out["d2h_GB_per_s_1_stream"], out["d2h_GB_per_s_2_streams"])
    per_file_d2h = 201 * 2 * 512 * 512 * 4 / (d2h * 1e9)
    per_file_write = 201 / out[f"writer_path_samples_per_s_{threads}_thread{'s' if threads > 1 else ''}"]
    per_file_indep = 201 * (indep["raw_write"] + indep["widen"]) / threads
    out.update({"threads": threads, "bound_s_per_file_d2h": per_file_d2h, "bound_s_per_file_host_write_and_widen": per_file_indep,
                "product_writer_path_s_per_file": per_file_write,
                "bound_files_per_s": 1.0 / max(per_file_d2h, per_file_indep),
                "basis": "per 10-minute file: 421 MB over the measured pinned D2H rate; 201 samples x (plain write of 4 MiB to a new file + "
                         "pwrite of 2 MiB + one float32 -> float64 pass; native loops, mg_host_io_probe), zero / random buffers, on the "
                         "writer threads' count of threads at once, in the record's scratch directory; the larger one.  "
                         "`product_writer_path_*`: the loop's own native call (mg_pt_write_samples) on the same threads, for comparison "
                         "-- not part of the bound"})
    return out


def stft_cpu_baseline(wav):
    """torch.stft on the host cores over the same 10-minute file (what the reference's torchaudio call lowers to)."""
    T = 1 + wav.numel() // 256
    torch.set_num_threads(host_cpu_share())
    w = wav.cpu()
    win = torch.hann_window(1024)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        c = torch.stft(w, 1024, 256, 1024, win, center=True, pad_mode="reflect", normalized=False, onesided=True,
                       return_complex=True)
        c = (c / win.pow(2.0).sum().sqrt())[:-1]
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": T / best, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "the same 10-minute file through torch.stft (what the reference's torchaudio call "
                      "lowers to, functions.py:53-62) + normalisation + Nyquist drop, min of 3"}


def host_cpu_share() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup cpu.max quota) -- the GPU box reports 256 logical CPUs
    but grants a 16-CPU quota, and oversubscribing that throttles the oracle by orders of magnitude."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(level: int, rand_channels: int, batch: int, iters: int):
    """The CPU oracle (plain-PyTorch port of train.py:135-221 as the reference executes it, G not detached in the D step)
    on the host cores of this box, bounded sample."""
    from oracle import progan as O
    torch.set_num_threads(host_cpu_share())
    torch.manual_seed(0)
    gs, ds = O.GenState(rand_channels), O.DiscState(7)
    for _ in range(level):
        gs.next_layer()
        ds.next_layer()
    side = LEVEL_SIDE[level]
    rng = torch.Generator().manual_seed(1234)
    x_real = torch.rand(batch, 2, side, side, generator=rng) * 2 - 1
    times = []
    for it in range(iters + 1):
        z = torch.randn(batch, rand_channels, 2, 2, generator=rng)
        z2 = torch.randn(batch, rand_channels, 2, 2, generator=rng)
        eps = torch.rand(batch, 1, 1, 1, generator=rng)
        t0 = time.perf_counter()
        O.d_step(gs, ds, x_real, z, eps, 0.5)
        O.g_step(gs, ds, z2, 0.5)
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
    best, med = min(times), sorted(times)[len(times) // 2]
    return {"value": batch / best, "median": batch / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"level {level} 2x{side}x{side}, batch {batch}, min (and median) of {iters} D+G steps after 1 warm-up "
                      f"(forward/backward only, reference-as-executed work incl. its non-detached D step)"}


def launch_ranks(n: int, argv) -> int:
    """`python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node n bench.py argv` as
    a child process with inherited stdout / stderr; returns its exit code.  MG_BENCH_LAUNCHER (a command line) replaces
    `python -m torch.distributed.run` -- the CPU test suite puts a recording stub there."""
    import shlex
    import subprocess
    launcher = os.environ.get("MG_BENCH_LAUNCHER")
    head = shlex.split(launcher) if launcher else [sys.executable, "-m", "torch.distributed.run"]
    # --standalone: the launcher binds its own rendezvous port on 127.0.0.1 (a port picked here by bind-and-close could be taken by
    # another bench run on the node before the launcher opens it)
    cmd = head + ["--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}", os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL between processes needs it on this image)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)    # >= 10 warm-up, >= 50 timed iterations
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--level", type=int, default=5, choices=[3, 4, 5, 6, 7])
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--rand-channels", type=int, default=32)
    ap.add_argument("--alpha", type=float, default=0.5,
                    help="fade-in coefficient of the timed step (SURVEY 8(d): 0.5 and 1.0; utils.py:62-68 holds 1.0 once a block has faded in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the other levels' records, the STFT / codec / create_dataset records and the train-loop record")
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--no-cadence", action="store_true",
                    help="skip the 5 critic : 1 generator `secondary` record (profiling runs: the trace then holds D+G steps only)")
    args = ap.parse_args()

    if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # dmabuf IPC (RCCL between processes needs it on this image): in THIS process, before the first HIP call -- a launcher that
        # starts `torch.distributed.run ... bench.py --gpus N` itself never passes through launch_ranks()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves -- as a CHILD process and before this
        # process has touched the GPU (an exec, or a fork after HIP initialisation, takes the node down on this pool) -- and pass
        # rank 0's JSON line and the exit code through.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)" +
                         (f" [rank {rank} of {world}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}]" if world > 1 else ""))
    # Rehearsal of the N > 1 code on a one-GPU box (tests/test_dp_gpu.py): MG_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # takes gloo, because RCCL refuses two ranks on one device.  The driver's launches never set it.
    share_gpu = os.environ.get("MG_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with a single rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train_step import ProGANStepper

    gen, disc = build_nets(args.level, args.rand_channels, device)
    optim_gen = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    optim_disc = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    side = LEVEL_SIDE[args.level]
    alpha = args.alpha
    data_rng = torch.Generator(device=device).manual_seed(1234 + rank)
    # latents and the penalty's epsilon: fresh per update from this seeded generator, drawn by the stepper as
    # train.py:143-149 / discriminator.py:166 do -- straight into the replayed graph's input buffers
    stepper = ProGANStepper(gen, disc, optim_gen, optim_disc, args.rand_channels, noise=data_rng)
    x_real = torch.rand(args.batch, 2, side, side, device=device, generator=data_rng) * 2 - 1

    def one_step():
        stepper.d_step(x_real, alpha)
        stepper.g_step(args.batch, alpha, device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # W untimed steps as asked -- and never fewer than four: an update runs eagerly twice and is captured as a HIP graph on its
    # third call (train_step.py), which must not fall into the timed region
    for _ in range(max(args.warmup, 4)):
        one_step()
    stepper.finish()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    stepper.finish()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen, rank_ms = 1, [1e3 * elapsed / args.steps] * 2  # [fastest, slowest] rank
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        dd = torch.distributed
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        tmin = t.clone()
        ones = torch.ones(1, dtype=torch.float32, device=device)
        dd.all_reduce(t, op=dd.ReduceOp.MAX)
        dd.all_reduce(tmin, op=dd.ReduceOp.MIN)
        dd.all_reduce(ones, op=dd.ReduceOp.SUM)  # a sum of ones over the data-path backend: the ranks that really took part
        elapsed = float(t.item())
        ranks_seen, rank_ms = int(round(float(ones.item()))), [1e3 * float(tmin.item()) / args.steps, 1e3 * elapsed / args.steps]
    ms_per_step = 1e3 * elapsed / args.steps
    images_per_s = world * args.batch * args.steps / elapsed

    # secondary figure: the reference's own cadence, five critic updates per generator update (train.py:189);
    # images/s counts the real images consumed (one batch per critic update).  Single GPU only.
    cadence = None
    if world == 1 and not args.no_cadence:
        def cycle():
            for _ in range(5):
                stepper.d_step(x_real, alpha)
            stepper.g_step(args.batch, alpha, device)
        cycle()
        stepper.finish()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        ncyc = max(1, args.steps // 5)
        for _ in range(ncyc):
            cycle()
        stepper.finish()
        torch.cuda.synchronize()
        tc = time.perf_counter() - tc
        cadence = {"cadence": "5 critic updates : 1 generator update (reference train.py:189)", "cycles": ncyc,
                   "ms_per_cycle": 1e3 * tc / ncyc, "images_per_s": 5 * args.batch * ncyc / tc}

    # BASELINE.json configs[4], the N-GPU half: STFT + codec preprocessing sharded by file (create_dataset.py: independent
    # units, no collective on the data path) -- every rank transforms its own synthetic 10-minute file between two barriers;
    # the whole-job figure is all ranks' frames over the slowest rank's time.
    stft_sharded = None
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    if dist_on and (world > 1 or os.environ.get("MG_FORCE_DP") == "1") and not args.no_extra:  # (1 rank: rehearsal only)
        from musicgan_amd import audio, ops
        wav = torch.rand(44100 * 600, device=device, generator=torch.Generator(device=device).manual_seed(7 + rank)) - 0.5
        frames = 1 + wav.numel() // 256
        for _ in range(4):
            audio.stft_to_phase_magn(ops.stft_1024(wav))
        torch.cuda.synchronize()
        barrier()
        reps = 10
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t1 = time.perf_counter()
        s0.record(torch.cuda.current_stream())
        for _ in range(reps):
            audio.stft_to_phase_magn(ops.stft_1024(wav))
        s1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
        if rank == 0:
            print(f"stft_sharded rank 0: {1e3 * wall / reps:.3f} ms/file wall, {s0.elapsed_time(s1) / reps:.3f} ms/file on the stream",
                  file=sys.stderr, flush=True)
        dt = torch.tensor([wall], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
        stft_sharded = {"workload": "one 10-minute 44.1 kHz file per rank through mg_stft_1024 + mg_codec_fwd, sharded by file",
                        "n_gpus": world, "frames_per_s": world * reps * frames / float(dt.item()),
                        "samples_per_s": world * reps * ((frames - 1) // 512) / float(dt.item()), "scaling": "weak"}

    if os.environ.get("MG_BENCH_CHECKSUM") and rank == 0:
        cs = sum(float(p.detach().double().abs().sum()) for p in list(gen.parameters()) + list(disc.parameters()))
        print(f"weights_abs_sum {cs:.10e}", file=sys.stderr, flush=True)
    if rank == 0:
        fpi = flops_per_image(args.level, args.rand_channels)
        xfpi = executed_flops_per_image(args.level, args.rand_channels, args.batch)
        dom = dominant_kernel_probe(device, args.batch)
        line = {
            "metric": f"spectrogram-images/sec G+D step, 2x{side}x{side} bs{args.batch}",
            "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "ranks_seen": ranks_seen, "ms_per_step_fastest_rank": rank_ms[0], "ms_per_step_slowest_rank": rank_ms[1],
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ProGAN level {args.level} WGAN-GP D+G step, 2x{side}x{side}, "
                                   f"batch {args.batch}/GPU, rand_channels {args.rand_channels}, alpha {args.alpha:g} (fade-in branch live), "
                                   f"Adam(1e-3,(0,0.9)) on both nets, random-init weights",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "roofline": {**step_roofline(fpi, xfpi, images_per_s / world),
                         # HBM bytes of ONE launch of the dominant kernel (per launch, like `dominant_kernel`), and of the whole step
                         "traffic": (dom.get("hbm_traffic") or {}).get("bytes_per_launch"),
                         "step_traffic": step_traffic(args.level, args.batch),
                         "traffic_basis": "NOT measured in this run: rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) read "
                                          "from profiles/traffic_dominant_kernel.json (tools/measure_traffic.sh) and "
                                          "profiles/traffic_step_l5.json (tools/measure_step_traffic.sh), re-measured in round 6",
                         # dominant launch, timed live with HIP events on its stream: frac = executed FLOPs over the peak
                         "dominant_kernel": {**dom, "frac": dom["executed_tflops"] / MFMA_F32_PEAK_TFLOPS,
                                             "algorithmic_frac": dom["tflops"] / MFMA_F32_PEAK_TFLOPS}},
        }
        if cadence is not None:
            line["secondary"] = cadence
        if stft_sharded is not None:
            line["stft_sharded"] = stft_sharded
        extra = world == 1 and not args.no_extra and args.level == 5
        if extra:
            # SURVEY 8(d)'s other single-GPU figures, timed in the same run: BASELINE.json configs[1] and configs[4].  Every GPU
            # measurement comes before the first CPU baseline: the oracle's OpenMP workers keep spinning for a while after their
            # last parallel region and, inside the box's 16-CPU quota, starve the launching thread (the 450 STFT launches then
            # run host-bound at 0.33 ms each instead of 0.12)
            # (l4: long enough to be past the transient that follows the level-5 run -- measured straight after it, the first 30
            # level-4 steps take 5.3 ms, a run on its own, 50 or 3 000 steps, 4.58)
            line["l4_bs32"] = level_record(device, args.rand_channels, 4, 32, 100, 60, "BASELINE.json configs[1]")
            # (levels 6-7 likewise: 30 steps straight after the previous record read 7.26 ms where a 160-step run on the same box gives 7.16)
            line["l6_bs6"] = level_record(device, args.rand_channels, 6, 6, 100, 40, "the reference's batch, train.py:43")
            line["l7_bs6"] = level_record(device, args.rand_channels, 7, 6, 80, 40, "the reference's batch, train.py:43")
            line["l7_bs16"] = level_record(device, args.rand_channels, 7, 16, 40, 20, "final level, larger batch")
            line["l3_bs8"] = level_record(device, args.rand_channels, 3, 8, 100, 30, "BASELINE.json configs[0], on the GPU")
            # SURVEY 8(d)'s second synthetic-input configuration: alpha = 1.0 (utils.py:62-68: where a stage sits once its block has
            # faded in; the fade-in branch still executes, generator.py:122-124 -- alpha is device data of the same captured graphs)
            line["alpha_1"] = level_record(device, args.rand_channels, 5, 64, 50, 20, "the headline configuration at alpha = 1.0", alpha=1.0)
            line["stft"] = stft_record(device, cpu=False)
            line.update(create_dataset_and_train_records(device, args.rand_channels))
        if world == 1 and not args.no_cpu_baseline:
            # bounded samples (~25 s in all): 1 warm-up + 1 timed step at level 5 (~9 s per step), 1 + 3 at levels 4 and 3 (< 1 s per
            # step; SURVEY 8(d): min of three, the median beside it)
            line["cpu_baseline"] = cpu_baseline(args.level, args.rand_channels, args.cpu_batch, iters=1 if args.level >= 5 else 3)
            if extra:
                line["l4_bs32"]["cpu_baseline"] = cpu_baseline(4, args.rand_channels, 32, iters=3)
                line["l3_bs8"]["cpu_baseline"] = cpu_baseline(3, args.rand_channels, 8, iters=3)
                wav = torch.rand(44100 * 600, device=device, generator=torch.Generator(device=device).manual_seed(7)) - 0.5
                line["stft"]["cpu_baseline"] = stft_cpu_baseline(wav)
        print(json.dumps(line), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
