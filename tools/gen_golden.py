"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference, container only).

Usage:  python tools/gen_golden.py            (writes tests/golden/progan_*.npz and audio_*.npz)

What is stored is data only: seeds, injected inputs, the reference's outputs, losses, gradients (full when small,
a fixed strided subsample + sum + L2 otherwise), post-Adam weights (same subsampling) and SHA-256 of every
reference-initialised parameter (so the oracle's same-seed init is checked bit-for-bit without storing 10 MB).
"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_loader import load_audio, load_networks, load_utils, wav_store  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from golden_util import c5_inverse_input, c5_sample_idx, c5_spectrum, c5_waveform  # noqa: E402  (the inputs the GPU box regenerates)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FULL_MAX = 4096
NSAMP = 509


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def sample_idx(numel: int) -> np.ndarray:
    if numel <= FULL_MAX:
        return np.arange(numel)
    return (np.arange(NSAMP, dtype=np.int64) * (numel // NSAMP))


def put_tensor(store, name, t):
    t = t.detach().to(torch.float32).contiguous().reshape(-1)
    idx = sample_idx(t.numel())
    store[name + "|samp"] = t.numpy()[idx]
    store[name + "|sum"] = np.float64(t.double().sum().item())
    store[name + "|l2"] = np.float64(t.double().norm().item())
    store[name + "|maxabs"] = np.float64(t.abs().max().item())


def progan_case(tag, seed, rand_channels, n_grow, alpha, batch, g_end_layer=0, d_start_layer=7, wscale=1.0,
                min_kink_margin=None):
    """Returns False (nothing written) when `min_kink_margin` is given and some LeakyReLU input of the step is closer to 0."""
    nets = load_networks()
    torch.manual_seed(seed)
    gen = nets.Generator(rand_channels, end_layer=g_end_layer)
    disc = nets.Discriminator(start_layer=d_start_layer)
    for _ in range(n_grow):
        gen.next_layer()
        disc.next_layer()
    if wscale != 1.0:
        # leave the near-zero-critic regime of a fresh init (|grad_x D| ~ 0 => GP gradients cancel): scale every
        # weight (not bias) of both nets in place; the oracle test applies the same scaling after its own init
        with torch.no_grad():
            for net in (gen, disc):
                for k, p in net.named_parameters():
                    if k.endswith("weight"):
                        p.mul_(wscale)
    store = {"seed": seed, "rand_channels": rand_channels, "n_grow": n_grow, "alpha": alpha, "batch": batch,
             "wscale": wscale,
             "g_end_layer": g_end_layer, "d_start_layer": d_start_layer,
             "g_curr_layer": gen.curr_layer, "d_curr_layer": disc.curr_layer}
    gsd, dsd = gen.state_dict(), disc.state_dict()
    store["g_keys"] = np.array(list(gsd.keys()))
    store["d_keys"] = np.array(list(dsd.keys()))
    store["g_shapes"] = np.array([str(tuple(v.shape)) for v in gsd.values()])
    store["d_shapes"] = np.array([str(tuple(v.shape)) for v in dsd.values()])
    store["g_sha"] = np.array([sha(v) for v in gsd.values()])
    store["d_sha"] = np.array([sha(v) for v in dsd.values()])

    side = 2 * 2 ** (gen.curr_layer + 1)
    rng = torch.Generator().manual_seed(seed + 1000)
    z = torch.randn(batch, rand_channels, 2, 2, generator=rng)
    z2 = torch.randn(batch, rand_channels, 2, 2, generator=rng)
    x_real = torch.rand(batch, 2, side, side, generator=rng) * 2 - 1
    eps = torch.rand(batch, 1, 1, 1, generator=rng)
    store.update(z=z.numpy(), z2=z2.numpy(), x_real=x_real.numpy(), eps=eps.numpy())

    # every LeakyReLU input of every forward pass of the step (D step: G(z), D(x_real), D(x_fake), D(x~); G step with the
    # updated critic: G(z2), D(x_fake2)) is watched: see kink_margin() / scan_seed()
    worst = [float("inf")]

    def _hook(_m, inp):
        x = inp[0].detach()
        worst[0] = min(worst[0], float(x.abs().min() / x.pow(2).mean().sqrt()))

    hooks = [m.register_forward_pre_hook(_hook) for net in (gen, disc) for m in net.modules()
             if isinstance(m, torch.nn.LeakyReLU)]

    optim_gen = torch.optim.Adam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    optim_disc = torch.optim.Adam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))

    # ---- D step, train.py:152-175 (eps injected by seeding the global RNG the way gradient_penalty draws it)
    x_fake = gen(z, alpha)
    out_real = disc(x_real, alpha)
    out_fake = disc(x_fake, alpha)
    d_loss = nets.wasserstein_discriminator_loss(out_real, out_fake)
    state = torch.get_rng_state()
    # gradient_penalty draws th.rand(batch,1,1,1) from the global generator: make that draw == eps
    torch.manual_seed(seed + 2000)
    eps_drawn = torch.rand(batch, 1, 1, 1)
    store["eps"] = eps_drawn.numpy()
    torch.manual_seed(seed + 2000)
    gp = disc.gradient_penalty(x_real, x_fake, alpha)
    torch.set_rng_state(state)
    gen.zero_grad()
    disc.zero_grad()
    (d_loss + gp).backward()
    store.update(x_fake=x_fake.detach().numpy(), out_real=out_real.detach().numpy(),
                 out_fake=out_fake.detach().numpy(), disc_loss=np.float64(d_loss.item()),
                 grad_pen=np.float64(gp.item()))
    live_d = [k for k, p in disc.named_parameters() if p.grad is not None]
    live_g = [k for k, p in gen.named_parameters() if p.grad is not None]
    store["dstep_d_live"] = np.array(live_d)
    store["dstep_g_live"] = np.array(live_g)
    for k, p in disc.named_parameters():
        if p.grad is not None:
            put_tensor(store, f"dstep_dgrad|{k}", p.grad)
    for k, p in gen.named_parameters():
        if p.grad is not None:
            put_tensor(store, f"dstep_ggrad|{k}", p.grad)
    optim_disc.step()
    for k, p in disc.named_parameters():
        put_tensor(store, f"dstep_dparam|{k}", p)

    # ---- G step, train.py:191-214 (with the updated discriminator)
    x_fake2 = gen(z2, alpha)
    out_fake2 = disc(x_fake2, alpha)
    g_loss = nets.wasserstein_generator_loss(out_fake2)
    gen.zero_grad()
    disc.zero_grad()
    g_loss.backward()
    store.update(x_fake2=x_fake2.detach().numpy(), out_fake2=out_fake2.detach().numpy(),
                 gen_loss=np.float64(g_loss.item()))
    for k, p in gen.named_parameters():
        if p.grad is not None:
            put_tensor(store, f"gstep_ggrad|{k}", p.grad)
    optim_gen.step()
    for k, p in gen.named_parameters():
        put_tensor(store, f"gstep_gparam|{k}", p)

    for h in hooks:
        h.remove()
    if min_kink_margin is not None:
        if worst[0] < min_kink_margin:
            return False
        store["kink_margin"] = worst[0]
    path = os.path.join(OUT, f"progan_{tag}.npz")
    np.savez_compressed(path, **store)
    print(f"wrote {path}: L{gen.curr_layer} side {side} d_loss {d_loss.item():.6f} gp {gp.item():.6f} "
          f"g_loss {g_loss.item():.6f}  ({os.path.getsize(path) / 1024:.0f} KiB)")
    return True



def kink_margin(nets_mod, gen, disc, alpha, z_list, x_list):
    """Smallest |LeakyReLU input| / rms(that layer's inputs) over G(z) for z in z_list and D(x) for x in x_list (x may be a callable
    taking the generated samples).  LeakyReLU's derivative jumps at 0: an input within fp32 round-off (~1e-6 of the layer's rms)
    of the kink makes every fp32 implementation's gradient a coin flip there, so fixtures keep clear of it."""
    worst = [float("inf")]

    def hook(_m, inp):
        x = inp[0].detach()
        worst[0] = min(worst[0], float(x.abs().min() / x.pow(2).mean().sqrt()))

    hs = [m.register_forward_pre_hook(hook) for net in (gen, disc) for m in net.modules()
          if isinstance(m, torch.nn.LeakyReLU)]
    with torch.no_grad():
        fakes = [gen(z, alpha) for z in z_list]
        for x in x_list:
            disc(x(fakes) if callable(x) else x, alpha)
        for f in fakes:
            disc(f, alpha)
    for h in hs:
        h.remove()
    return worst[0]


def _scale_weights(nets_list, wscale):
    with torch.no_grad():
        for net in nets_list:
            for k, p in net.named_parameters():
                if k.endswith("weight"):
                    p.mul_(wscale)


def find_wscale(seed, rand_channels, n_grow, alpha, batch, target_norm):
    """Weight scale (bisection, 3 significant digits) at which the MEDIAN per-sample ||grad_x D(x~)|| of the reference critic
    equals `target_norm` on this case's inputs -- leaves the near-zero-critic regime of a fresh init, where the penalty sits at
    10 and its gradient is a cancellation residue."""
    nets = load_networks()
    torch.manual_seed(seed)
    gen, disc = nets.Generator(rand_channels), nets.Discriminator(7)
    for _ in range(n_grow):
        gen.next_layer()
        disc.next_layer()
    base = [p.detach().clone() for net in (gen, disc) for p in net.parameters()]
    side = 2 * 2 ** (gen.curr_layer + 1)
    rng = torch.Generator().manual_seed(seed + 1000)
    z = torch.randn(batch, rand_channels, 2, 2, generator=rng)
    torch.randn(batch, rand_channels, 2, 2, generator=rng)
    x_real = torch.rand(batch, 2, side, side, generator=rng) * 2 - 1
    torch.manual_seed(seed + 2000)
    eps = torch.rand(batch, 1, 1, 1)

    def med_norm(ws):
        with torch.no_grad():
            for p, b in zip([p for net in (gen, disc) for p in net.parameters()], base):
                p.copy_(b * ws if p.dim() > 1 else b)
            x_fake = gen(z, alpha)
        xi = (eps * x_real + (1 - eps) * x_fake).requires_grad_(True)
        (g,) = torch.autograd.grad(disc(xi, alpha).sum(), xi)
        return float(g.reshape(batch, -1).norm(dim=1).median())

    lo, hi = 1.0, 8.0
    for _ in range(14):
        mid = 0.5 * (lo + hi)
        if med_norm(mid) < target_norm:
            lo = mid
        else:
            hi = mid
    return float(f"{0.5 * (lo + hi):.3g}")


def trajectory_case(tag, seed, rand_channels, batch, iters, fadein, train_lengths, wscale=1.0):
    """The reference's training loop body (train.py:131-272) for `iters` loader batches on the reference's own Generator /
    Discriminator / Grower and torch.optim.Adam: D step every iteration, G step when iter_idx % 5 == 0, growth through
    Grower.grow + next_layer + add_param_group.  Only the device (CPU), the injected inputs (z, x_real at the level's size, the
    RNG seed in front of gradient_penalty's eps draw and of next_layer's fresh head/stem) and the dtype differ from train.py.

    Run twice: float64 (the yard-stick the oracle is pinned on: deterministic to ~1e-12, so the whole trajectory can be
    compared) and float32 (as the reference really runs; stored to document that the reference's own fp32 trajectory leaves
    its fp64 one within a few iterations -- Adam(beta1=0) moves every weight by ~lr*sign(g), so round-off in near-zero gradient
    entries is amplified to O(lr) per step and the GAN dynamics do the rest)."""
    nets, U = load_networks(), load_utils()
    betas = (0.0, 0.9)

    def run(dtype):
        torch.manual_seed(seed)
        gen, disc = nets.Generator(rand_channels, end_layer=0), nets.Discriminator(start_layer=7)
        init_sha = ([sha(v) for v in gen.state_dict().values()], [sha(v) for v in disc.state_dict().values()])
        if wscale != 1.0:
            _scale_weights((gen, disc), wscale)
        gen, disc = gen.to(dtype), disc.to(dtype)
        og = torch.optim.Adam(gen.parameters(), lr=1e-3, betas=betas)
        od = torch.optim.Adam(disc.parameters(), lr=1e-3, betas=betas)
        grower = U.Grower(n_grow=7, fadein_lengths=list(fadein), train_lengths=list(train_lengths))
        rng = torch.Generator().manual_seed(seed + 1000)
        recs, inputs, heads = [], [], []
        for it in range(iters):
            side = 4 * 2 ** gen.curr_layer
            x_real = torch.rand(batch, 2, side, side, generator=rng) * 2 - 1
            z = torch.randn(batch, rand_channels, 2, 2, generator=rng)
            z2 = torch.randn(batch, rand_channels, 2, 2, generator=rng)
            alpha = grower.alpha
            x_fake = gen(z.to(dtype), alpha)
            out_real, out_fake = disc(x_real.to(dtype), alpha), disc(x_fake, alpha)
            d_loss = nets.wasserstein_discriminator_loss(out_real, out_fake)
            torch.manual_seed(seed + 2000 + it)  # gradient_penalty draws eps = th.rand(batch,1,1,1) from the global generator
            eps = torch.rand(batch, 1, 1, 1)
            torch.manual_seed(seed + 2000 + it)
            gp = disc.gradient_penalty(x_real.to(dtype), x_fake, alpha)
            gen.zero_grad()
            disc.zero_grad()
            (d_loss + gp).backward()
            od.step()
            r = {"level": gen.curr_layer, "alpha": alpha, "disc_loss": d_loss.item(), "grad_pen": gp.item(),
                 "out_real": out_real.mean().item(), "out_fake": out_fake.mean().item(), "gen_loss": np.nan, "grew": 0}
            if it % 5 == 0:
                x_fake = gen(z2.to(dtype), alpha)
                out_fake = disc(x_fake, alpha)
                g_loss = nets.wasserstein_generator_loss(out_fake)
                gen.zero_grad()
                disc.zero_grad()
                g_loss.backward()
                og.step()
                r["gen_loss"] = g_loss.item()
            inputs.append((x_real, z, z2, eps))
            if grower.grow(batch) and gen.growing:
                torch.manual_seed(seed + 3000 + gen.curr_layer)  # fresh head / stem come from the global generator
                gen.next_layer()
                disc.next_layer()
                heads.append([sha(p) for p in gen.end_block_params()] + [sha(p) for p in disc.start_block_parameters()])
                gen, disc = gen.to(dtype), disc.to(dtype)
                og.add_param_group({"params": gen.end_block_params(), "lr": 1e-3, "betas": betas})
                od.add_param_group({"params": disc.start_block_parameters(), "lr": 1e-3, "betas": betas})
                r["grew"] = 1
            recs.append(r)
        return recs, inputs, heads, gen, disc, og, od, init_sha

    r64, inputs, heads, gen, disc, og, od, init_sha = run(torch.float64)
    r32, _, _, gen32, disc32 = run(torch.float32)[:5]
    store = {"seed": seed, "rand_channels": rand_channels, "batch": batch, "iters": iters, "wscale": wscale,
             "fadein": np.array(fadein), "train_lengths": np.array(train_lengths),
             "g_init_sha": np.array(init_sha[0]), "d_init_sha": np.array(init_sha[1]), "new_head_sha": np.array(heads)}
    for it, (x_real, z, z2, eps) in enumerate(inputs):
        store.update({f"x_real|{it}": x_real.numpy(), f"z|{it}": z.numpy(), f"z2|{it}": z2.numpy(), f"eps|{it}": eps.numpy()})
    for name, recs in (("ref64", r64), ("ref32", r32)):
        for key in ("level", "alpha", "disc_loss", "grad_pen", "out_real", "out_fake", "gen_loss", "grew"):
            store[f"{name}|{key}"] = np.array([r[key] for r in recs], dtype=np.float64)
    # final state of the float64 run: every parameter (subsampled), and the Adam state the loop built up
    store["g_keys"] = np.array(list(gen.state_dict().keys()))
    store["d_keys"] = np.array(list(disc.state_dict().keys()))
    for pre, net, opt in (("g", gen, og), ("d", disc, od)):
        steps = {}
        for k, p in net.named_parameters():
            store[f"final64|{pre}|{k}|samp"] = p.detach().reshape(-1).numpy()[sample_idx(p.numel())]
            st = opt.state.get(p)
            steps[k] = int(st["step"]) if st else 0
            if st:
                store[f"final64|{pre}|{k}|exp_avg_sq|samp"] = st["exp_avg_sq"].reshape(-1).numpy()[sample_idx(p.numel())]
        for k, p in (gen32 if pre == "g" else disc32).named_parameters():  # the float32 run's final weights, for the noise scale
            store[f"final32|{pre}|{k}|samp"] = p.detach().reshape(-1).numpy()[sample_idx(p.numel())]
        store[f"adam_steps|{pre}|keys"] = np.array(list(steps.keys()))
        store[f"adam_steps|{pre}"] = np.array(list(steps.values()))
        store[f"adam_groups|{pre}"] = np.array([len(gr["params"]) for gr in opt.param_groups])
    path = os.path.join(OUT, f"progan_{tag}.npz")
    np.savez_compressed(path, **store)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB): levels {[int(r['level']) for r in r64]}")
    for a, b in zip(r64, r32):
        print(f"   L{a['level']} alpha {a['alpha']:.3f}  d_loss {a['disc_loss']:+.6e} (fp32 off by {abs(a['disc_loss'] - b['disc_loss']):.1e})"
              f"  gp {a['grad_pen']:.6f} (fp32 off by {abs(a['grad_pen'] - b['grad_pen']):.1e})")


def transforms_case():
    """audio/transforms.py:4-40 (ChannelMinMaxNorm, ChangeRange) run by the reference itself on float32 and on float64-born
    samples the way train.py:139 feeds them (`.to(th.float)` first), plus the reference Grower's whole scale_transform at two
    levels.  The Resize inside scale_transform is the torchvision stand-in of tools/ref_loader.py (aten bilinear + antialias):
    `scaled_*` is therefore pinned on torch, `norm_*` / `ranged_*` on the reference's own code alone."""
    audio, U = load_audio(), load_utils()
    rng = torch.Generator().manual_seed(31)
    x = torch.randn(3, 2, 64, 64, generator=rng, dtype=torch.float64) * torch.tensor([1.0, 40.0]).view(1, 2, 1, 1) + 3.0
    x[1, 0] = 0.25  # a constant channel: (x - min) / (0 + eps)
    xf = x.to(torch.float)
    norm = audio.ChannelMinMaxNorm()(xf)
    ranged = audio.ChangeRange(-1.0, 1.0)(norm)
    store = {"x64": x.numpy(), "norm": norm.numpy(), "ranged": ranged.numpy()}
    # one stored sample (2,512,512) as create_dataset writes it (float64 on disk; values kept float32-representable so the
    # fixture stores 2 MB instead of 4), through the reference Grower's scale_transform at levels 0, 3 and 5
    big = torch.rand(1, 2, 512, 512, generator=rng) * 2 - 1
    big[:, 1] = big[:, 1].cumsum(-1) * 0.01
    store["big32"] = big.numpy()
    grower = U.Grower(n_grow=7, fadein_lengths=[1] * 8, train_lengths=[1] * 7)
    for level in range(6):
        if level in (0, 3, 5):
            store[f"scaled_l{level}"] = grower.scale_transform(big.to(torch.float64).to(torch.float)).numpy()
        grower.grow(2)
    path = os.path.join(OUT, "transforms.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, f"({os.path.getsize(path) / 1024:.0f} KiB)")


def progan_shapes_case():
    """The walk of networks/test_networks.py:4-38 (shapes at every level, growing flags), plus a non-square
    forward as generate.py:47-54 uses it."""
    nets = load_networks()
    torch.manual_seed(5)
    gen, disc = nets.Generator(8), nets.Discriminator(7)
    shapes, dshapes, flags = [], [], []
    for _ in range(gen.down_sample + 3):
        z = torch.randn(1, 8, 2, 2)
        with torch.no_grad():
            out = gen(z, 0.5)
            dout = disc(out, 0.5)
        shapes.append(list(out.shape))
        dshapes.append(list(dout.shape))
        flags.append([int(gen.growing), int(disc.growing)])
        gen.next_layer()
        disc.next_layer()
    store = {"g_out_shapes": np.array(shapes), "d_out_shapes": np.array(dshapes), "growing": np.array(flags),
             "g_keys_final": np.array(list(gen.state_dict().keys())),
             "d_keys_final": np.array(list(disc.state_dict().keys()))}
    # non-square generator forward, directly constructed at end_layer=2 (fresh previous head, generator.py:93-104)
    torch.manual_seed(6)
    g2 = nets.Generator(8, end_layer=2)
    z = torch.randn(2, 8, 2, 6, generator=torch.Generator().manual_seed(66))
    with torch.no_grad():
        y = g2(z, 1.0)
        y37 = g2(z, 0.37)
    store.update(ns_z=z.numpy(), ns_out=y.numpy(), ns_out_a037=y37.numpy(),
                 ns_keys=np.array(list(g2.state_dict().keys())),
                 ns_sha=np.array([sha(v) for v in g2.state_dict().values()]))
    path = os.path.join(OUT, "progan_shapes.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, shapes[:3], "...", shapes[-1])


def audio_case():
    audio = load_audio()
    fn = sys.modules["music_gan.audio.functions"]
    rng = torch.Generator().manual_seed(7)
    sr = 44100
    # 3.2 s stereo: 552 frames -> one 512-frame sample after dropping the first frame and the remainder
    wav = (torch.rand(2, 141_312, generator=rng) - 0.5)
    t = torch.arange(wav.shape[1]) / sr
    wav[0] += 0.3 * torch.sin(2 * np.pi * 440.0 * t)
    wav[1] += 0.2 * torch.sin(2 * np.pi * 3000.0 * t + 1.0)
    wav_store()["in.wav"] = (wav, sr)
    c = audio.wav_to_stft("in.wav")
    magn, phase = audio.stft_to_phase_magn(c, nb_vec=512)
    store = {"wav": wav.numpy(), "stft_real": c.real.numpy().astype(np.float32),
             "stft_imag": c.imag.numpy().astype(np.float32), "magn": magn.numpy(), "phase": phase.numpy()}
    # bark scale vector on ones (functions.py:26-35)
    store["bark_scale"] = audio.bark_magn_scale(torch.ones(512, 1)).numpy()[:, 0]
    # unwrap on random phases (functions.py:17-23)
    ph = (torch.rand(16, 300, generator=rng) * 2 - 1) * np.pi
    store["unwrap_in"] = ph.numpy()
    store["unwrap_out"] = fn.unwrap(ph.clone()).numpy()
    # inverse path (functions.py:97-139) on the codec output, restricted to the first 64 frames
    mp = torch.stack([magn[:, :, :64], phase[:, :, :64]], dim=1)  # (1,2,512,64)
    audio.magn_phase_to_wav(mp.clone(), "out.wav", sr)
    out_wav, _ = wav_store()["out.wav"]
    store["inv_in"] = mp.numpy()
    store["inv_wav"] = out_wav.numpy()
    path = os.path.join(OUT, "audio_codec.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, "stft", tuple(c.shape), "magn", tuple(magn.shape), "wav out", tuple(out_wav.shape),
          f"({os.path.getsize(path) / 1024:.0f} KiB)")
    # notebook cell 7 known answer: 30 s mono -> [513, 5168] before the Nyquist drop
    wav_store()["k.wav"] = (torch.zeros(1, 44100 * 30), sr)
    assert tuple(audio.wav_to_stft("k.wav").shape) == (512, 5168)


def c5_put(store, name, t: torch.Tensor):
    """Strided subsample + sum + l2 + maxabs, plus full rows (bins 0, 255, 511) of the first and the last image."""
    flat = t.detach().contiguous().reshape(-1)
    store[name + "|samp"] = flat.numpy()[c5_sample_idx(flat.numel())]
    store[name + "|sum"] = np.float64(flat.double().sum().item())
    store[name + "|l2"] = np.float64(flat.double().norm().item())
    store[name + "|maxabs"] = np.float64(flat.abs().max().item())
    if t.dim() == 3:
        store[name + "|rows"] = t[[0, 0, 0, -1, -1, -1], [0, 255, 511, 0, 255, 511], :].numpy()


def audio_config5_case():
    """BASELINE config 5 at its stated size (create_dataset.py:34-64 -> functions.py:38-94): the reference's own wav_to_stft +
    stft_to_phase_magn on a 10-minute track (103 360 frames -> 201 images), its stft_to_phase_magn on a library-independent
    complex input of the same size, and its magn_phase_to_wav (functions.py:97-139) over 20 480 frames."""
    audio = load_audio()
    fn = sys.modules["music_gan.audio.functions"]
    sr = 44100
    store = {}
    # (1) waveform -> STFT -> codec
    wav = c5_waveform()
    store["wav|sha256"] = hashlib.sha256(wav.tobytes()).hexdigest()
    wav_store()["c5.wav"] = (torch.from_numpy(wav)[None, :], sr)
    c = audio.wav_to_stft("c5.wav")
    assert tuple(c.shape) == (512, 103360)
    store["wav|stft_sha256"] = sha(torch.view_as_real(c))
    store["wav|stft_maxabs"] = np.float64(c.abs().max().item())
    cs = torch.view_as_real(c).reshape(-1)
    store["wav|stft_samp"] = cs.numpy()[c5_sample_idx(cs.numel())]
    magn, phase = audio.stft_to_phase_magn(c, nb_vec=512)
    assert tuple(magn.shape) == tuple(phase.shape) == (201, 512, 512)
    c5_put(store, "wav|magn", magn)
    c5_put(store, "wav|phase", phase)
    u = fn.unwrap(torch.angle(c))
    d = u[:, 1:] - u[:, :-1]
    store["wav|unwrapped_maxabs"] = np.float64(u.abs().max().item())
    store["wav|delta_min"], store["wav|delta_max"] = np.float32(d.min().item()), np.float32(d.max().item())
    del c, cs, magn, phase, u, d
    # (2) library-independent complex input -> codec
    x = torch.from_numpy(c5_spectrum())
    store["spec|sha256"] = sha(torch.view_as_real(x))
    magn, phase = audio.stft_to_phase_magn(x, nb_vec=512)
    c5_put(store, "spec|magn", magn)
    c5_put(store, "spec|phase", phase)
    u = fn.unwrap(torch.angle(x))
    d = u[:, 1:] - u[:, :-1]
    store["spec|unwrapped_maxabs"] = np.float64(u.abs().max().item())
    store["spec|delta_min"], store["spec|delta_max"] = np.float32(d.min().item()), np.float32(d.max().item())
    del x, magn, phase, u, d
    # (3) inverse over 20 480 frames
    mp = torch.from_numpy(c5_inverse_input())
    store["inv|sha256"] = sha(mp)
    audio.magn_phase_to_wav(mp.clone(), "c5_out.wav", sr)
    out, _ = wav_store()["c5_out.wav"]
    out = out.reshape(-1)
    assert out.numel() == 256 * (20480 - 1)
    c5_put(store, "inv|wav", out)
    store["inv|wav|head"], store["inv|wav|tail"] = out[:4096].numpy(), out[-4096:].numpy()
    path = os.path.join(OUT, "audio_config5.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, f"({os.path.getsize(path) / 1024:.0f} KiB)")


KINK_MARGIN = 3e-6


def scan_seed(tag, seed0, target_norm=None, **kw):
    """First seed >= seed0 whose step keeps every LeakyReLU input at least KINK_MARGIN (in units of that layer's rms) away from
    0.  LeakyReLU's derivative jumps at 0, so an input within fp32 round-off of the kink (the HIP kernels and the CPU library both
    sit at 2e-7 .. 1e-6 rms, tools/diag_act_noise.py) makes the gradient of ANY fp32 implementation differ from fp64 by one
    flipped mask element -- 1e-3 of a tensor on these small maps; roughly one seed in 80 keeps all ~2e6 LeakyReLU inputs of the
    D step and the G step clear of it by 3e-6."""
    for seed in range(seed0, seed0 + 4000):
        ws = kw.get("wscale", 1.0)
        if target_norm is not None:
            ws = find_wscale(seed=seed, rand_channels=kw["rand_channels"], n_grow=kw["n_grow"], alpha=kw["alpha"],
                             batch=kw["batch"], target_norm=target_norm)
        if progan_case(tag, seed=seed, min_kink_margin=KINK_MARGIN, **{**kw, "wscale": ws}):
            return seed
    raise RuntimeError(f"no seed found for {tag}")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--config5-only" in sys.argv:
        audio_config5_case()
        sys.exit(0)
    scan_seed("l0_rc8", 11, rand_channels=8, n_grow=0, alpha=1.0, batch=3)
    scan_seed("l1_rc8_fade", 12, rand_channels=8, n_grow=1, alpha=0.37, batch=3)
    scan_seed("l3_rc32_fade", 13, rand_channels=32, n_grow=3, alpha=0.37, batch=2)
    scan_seed("l2_direct", 14, rand_channels=16, n_grow=0, alpha=0.6, batch=2, g_end_layer=2, d_start_layer=5)
    scan_seed("l2_rc16_fade_scaled", 15, rand_channels=16, n_grow=2, alpha=0.5, batch=4, wscale=1.7)
    # the well-conditioned penalty regime: weights scaled until the median ||grad_x D(x~)|| is ~1 (samples on both sides of 1,
    # so (||g|| - 1) takes both signs) and ~3 (all positive)
    scan_seed("l2_rc16_gpnorm1", 16, target_norm=1.0, rand_channels=16, n_grow=2, alpha=0.5, batch=4)
    scan_seed("l2_rc16_gpnorm3", 16, target_norm=3.0, rand_channels=16, n_grow=2, alpha=0.5, batch=4)
    trajectory_case("trajectory", seed=21, rand_channels=8, batch=3, iters=16, fadein=[1, 12, 12, 12, 12, 12, 12, 12],
                    train_lengths=[15, 15, 15, 15, 15, 15, 15])
    transforms_case()
    progan_shapes_case()
    audio_case()
    audio_config5_case()
