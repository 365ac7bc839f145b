"""GPU parity of the audio path (STFT, codec, inverse) through the reference-named functions against golden vectors
captured from the reference's audio/functions.py and against the numpy oracle; plus the create_dataset / train / generate
drivers end to end on a tiny synthetic corpus."""
import os

import numpy as np
import pytest
import torch

from golden_util import load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_wav_to_stft_and_codec_match_reference_golden(tmp_path):
    from musicgan_amd import audio
    from musicgan_amd.audio import wavio
    g = load("audio_codec.npz")
    path = str(tmp_path / "in.wav")
    wavio.save(path, torch.from_numpy(g["wav"]), 44100)
    c = audio.wav_to_stft(path)
    ref = torch.from_numpy(g["stft_real"] + 1j * g["stft_imag"])
    assert c.is_cuda and tuple(c.shape) == (512, 553)
    assert float((c.cpu() - ref).abs().max() / ref.abs().max()) <= 1e-5
    magn, phase = audio.stft_to_phase_magn(c)
    assert tuple(magn.shape) == tuple(phase.shape) == (1, 512, 512)
    assert float((magn.cpu() - torch.from_numpy(g["magn"])).abs().max()) <= 5e-6
    # atan2 + a 553-step fp32 running sum: 1e-4 of the [-1, 1] range (the oracle test uses the same bound)
    assert float((phase.cpu() - torch.from_numpy(g["phase"])).abs().max()) <= 1e-4
    # codec on the REFERENCE's own complex values isolates the codec kernel from the STFT kernel
    magn2, phase2 = audio.stft_to_phase_magn(ref.to(torch.complex64))
    assert float((magn2.cpu() - torch.from_numpy(g["magn"])).abs().max()) <= 2e-6
    assert float((phase2.cpu() - torch.from_numpy(g["phase"])).abs().max()) <= 1e-4
    s = audio.bark_magn_scale(torch.ones(512, 1, device=DEV))[:, 0].cpu().numpy()
    assert np.allclose(s, g["bark_scale"], rtol=1e-6, atol=0)
    with pytest.raises(AssertionError):
        audio.bark_magn_scale(torch.ones(4, 4, 4, device=DEV))


def test_inverse_codec_matches_reference_golden(tmp_path):
    from musicgan_amd import audio
    from musicgan_amd.audio import wavio
    from oracle import audio as OA
    g = load("audio_codec.npz")
    mp = torch.from_numpy(g["inv_in"]).to(DEV)
    wav = audio.magn_phase_to_waveform(mp).cpu().numpy()
    ref = g["inv_wav"].reshape(-1)
    assert wav.shape == ref.shape == (256 * 63,)
    assert float(np.abs(wav - ref).max()) <= 2e-3 * float(np.abs(ref).max())
    assert float(np.abs(wav - OA.magn_phase_to_wav(g["inv_in"])).max()) <= 2e-3 * float(np.abs(ref).max())
    out = str(tmp_path / "o.wav")
    audio.magn_phase_to_wav(mp, out, 44100)
    back, sr = wavio.load(out)
    assert sr == 44100 and tuple(back.shape) == (1, 256 * 63)
    # two items concatenate along time (functions.py:108-109)
    mp2 = torch.cat([mp, mp], dim=0)
    assert audio.magn_phase_to_waveform(mp2).numel() == 256 * (2 * 64 - 1)
    for bad in (torch.zeros(2, 512, 4), torch.zeros(1, 3, 512, 4), torch.zeros(1, 2, 100, 4)):
        with pytest.raises(AssertionError):
            audio.magn_phase_to_waveform(bad.to(DEV))


def _c5(case):
    import hashlib
    from golden_util import c5_inverse_input, c5_spectrum, c5_waveform
    g = load("audio_config5.npz")
    x = {"wav": c5_waveform, "spec": c5_spectrum, "inv": c5_inverse_input}[case]()
    key = "wav|sha256" if case == "wav" else f"{case}|sha256"
    assert hashlib.sha256(np.ascontiguousarray(x).view(np.float32).tobytes()).hexdigest() == str(g[key]), \
        "the regenerated config-5 input is not the one the reference was given"
    return g, x


def _c5_check_magn(g, case, magn, tol):
    from golden_util import c5_sample_idx
    got = magn.reshape(-1)[torch.from_numpy(c5_sample_idx(magn.numel())).to(magn.device)].cpu().numpy()
    assert float(np.abs(got - g[f"{case}|magn|samp"]).max()) <= tol
    rows = magn[[0, 0, 0, -1, -1, -1], [0, 255, 511, 0, 255, 511], :].cpu().numpy()
    assert float(np.abs(rows - g[f"{case}|magn|rows"]).max()) <= tol
    assert abs(float(magn.double().sum()) - float(g[f"{case}|magn|sum"])) <= tol * magn.numel() * 0.05


def test_codec_at_config5_size_against_the_reference():
    """BASELINE config 5 at its stated size: stft_to_phase_magn (functions.py:65-94) on 103 360 frames -> 201 images against the
    REFERENCE's output on the same bits (library-independent STFT-like input whose unwrapped phase reaches 1.6e5 rad).
    The phase image is discontinuous in angle(X): one ulp of atan2f moves the float32 rounding of the running sum, which is why
    rounds 1-3 (rocm's atan2f) matched the reference on 94 % of the elements only.  The kernel now evaluates th.abs / th.angle
    with the algorithm of the library torch calls (csrc/sleef_f32.h) and keeps torch.cumsum's float64 accumulator, so EVERY sampled
    element of both images must be the reference's to the bit (a float32 running sum -- rounds 1-2 -- sat at 0.05 %)."""
    from golden_util import c5_phase_stats, c5_sample_idx
    from musicgan_amd import audio
    from oracle import audio as OA
    g, x = _c5("spec")
    xd = torch.from_numpy(x).to(DEV)
    magn, phase = audio.stft_to_phase_magn(xd)
    assert tuple(magn.shape) == tuple(phase.shape) == (201, 512, 512)
    frac, worst, flips, n = c5_phase_stats(g, "spec", phase.cpu().numpy())
    print(f"config-5 codec vs reference: {frac:.4%} of {n} within 1e-6, worst {worst:.3f} of the 2-ulp bound, {flips} flips")
    assert frac == 1.0 and worst == 0.0 and flips == 0, (frac, worst, flips, n)
    idx = torch.from_numpy(c5_sample_idx(phase.numel())).to(DEV)
    assert np.array_equal(phase.reshape(-1)[idx].cpu().numpy(), g["spec|phase|samp"]), "phase samples are not the reference's bits"
    _c5_check_magn(g, "spec", magn, 0.0)
    assert float(magn.min()) == -1.0 and float(magn.max()) == 1.0 and float(phase.min()) == -1.0 and float(phase.max()) == 1.0
    # the stacked form create_dataset uses is the same bits in one (S, 2, 512, 512) tensor
    from musicgan_amd import ops
    both = ops.codec_fwd(xd, audio.functions._bark_vector(512, xd.device), 512, stacked=True)
    assert torch.equal(both[:, 0], magn) and torch.equal(both[:, 1], phase)
    del both
    # and the whole of both images, not only the fixture's samples, against the oracle with torch's CPU abs / angle
    m_or, p_or = OA.stft_to_phase_magn(x, lib="torch")
    assert np.array_equal(p_or, phase.cpu().numpy()) and np.array_equal(m_or, magn.cpu().numpy())


def test_waveform_to_codec_at_config5_size_against_the_reference():
    """The seed-7 10-minute track of SURVEY 8(d) through mg_stft_1024 + mg_codec_fwd against the reference's wav_to_stft +
    stft_to_phase_magn (create_dataset.py:34-64).  The bins agree to 1e-5 of max|X| (float32 FFTs of different factorisations),
    so weak bins' phases differ by more than an ulp and the phase bound is the distributional one with a lower share of
    1e-6-exact elements (the oracle's float64 FFT: 92 %)."""
    from golden_util import c5_phase_stats, c5_sample_idx
    from musicgan_amd import audio
    g, wav = _c5("wav")
    c = audio.functions.stft_from_waveform(torch.from_numpy(wav).to(DEV))
    assert tuple(c.shape) == (512, 103360)
    cs = torch.view_as_real(c).reshape(-1)[torch.from_numpy(c5_sample_idx(2 * c.numel())).to(DEV)].cpu().numpy()
    assert float(np.abs(cs - g["wav|stft_samp"]).max()) <= 1e-5 * float(g["wav|stft_maxabs"])
    magn, phase = audio.stft_to_phase_magn(c)
    frac, worst, flips, n = c5_phase_stats(g, "wav", phase.cpu().numpy())
    print(f"config-5 wav -> codec vs reference: {frac:.4%} of {n} within 1e-6, worst {worst:.3f} of the 2-ulp bound, {flips} flips")
    assert frac >= 0.90 and worst <= 1.0 and flips <= 3, (frac, worst, flips, n)  # measured 92.1 %
    _c5_check_magn(g, "wav", magn, 5e-6)


def test_inverse_codec_over_20480_frames_against_the_reference():
    """magn_phase_to_wav (functions.py:97-139) on 40 images = 20 480 frames, the length `generate` runs it at."""
    from golden_util import c5_sample_idx
    from musicgan_amd import audio
    g, mp = _c5("inv")
    wav = audio.magn_phase_to_waveform(torch.from_numpy(mp).to(DEV))
    assert wav.numel() == 256 * (20480 - 1)
    scale = float(g["inv|wav|maxabs"])
    idx = torch.from_numpy(c5_sample_idx(wav.numel())).to(DEV)
    errs = [float(np.abs(wav[idx].cpu().numpy() - g["inv|wav|samp"]).max()),
            float(np.abs(wav[:4096].cpu().numpy() - g["inv|wav|head"]).max()),
            float(np.abs(wav[-4096:].cpu().numpy() - g["inv|wav|tail"]).max())]
    print("inverse over 20 480 frames: max error / max|wav| =", [e / scale for e in errs])
    assert max(errs) <= 1e-5 * scale, errs  # measured 2e-7: the cumulative phase is the same sequential float32 sum


def test_stft_from_pcm_frames_equals_the_normalised_mono_path(tmp_path):
    """mg_stft_1024_pcm: a file's frames as stored (int16 / float32 / int32 / uint8, 1-3 interleaved channels) -> the bins of
    `th_audio.load` (normalised to [-1, 1]) + `mean(0)` + spectrogram (functions.py:43-62).  float32 and int16 with one or two
    channels are converted inside the STFT kernel's loads: bit-identical to the mono float32 path; the other formats take one
    conversion pass: same bits again.  Also through a real stereo int16 file and `wav_to_stft(path)`."""
    from scipy.io import wavfile
    from musicgan_amd import audio, ops
    from musicgan_amd.audio import wavio
    rng = np.random.default_rng(21)
    frames = 256 * 40 + 77
    cases = {
        "f32x1": (rng.random((frames, 1), dtype=np.float32) - 0.5),
        "f32x2": (rng.random((frames, 2), dtype=np.float32) - 0.5),
        "i16x1": rng.integers(-32768, 32767, (frames, 1), dtype=np.int16),
        "i16x2": rng.integers(-32768, 32767, (frames, 2), dtype=np.int16),
        "i32x2": rng.integers(-2 ** 31, 2 ** 31 - 1, (frames, 2), dtype=np.int32),
        "u8x1": rng.integers(0, 255, (frames, 1), dtype=np.uint8),
        "f32x3": (rng.random((frames, 3), dtype=np.float32) - 0.5),
    }
    for name, pcm in cases.items():
        if pcm.dtype == np.int16:
            x = pcm.astype(np.float32) / 32768.0
        elif pcm.dtype == np.int32:
            x = pcm.astype(np.float32) / 2147483648.0
        elif pcm.dtype == np.uint8:
            x = (pcm.astype(np.float32) - 128.0) / 128.0
        else:
            x = pcm
        mono = torch.from_numpy(np.ascontiguousarray(x.T)).mean(0)          # the reference's order: normalise, then mean(0)
        want = ops.stft_1024(mono.to(DEV).contiguous())
        got = ops.stft_1024_pcm(torch.from_numpy(pcm).to(DEV))
        assert torch.equal(torch.view_as_real(got), torch.view_as_real(want)), name
    path = str(tmp_path / "stereo16.wav")
    wavfile.write(path, 44100, cases["i16x2"])
    loaded, sr = wavio.load(path)
    assert sr == 44100 and tuple(loaded.shape) == (2, frames)
    want = ops.stft_1024(loaded.mean(0).to(DEV).contiguous())
    assert torch.equal(torch.view_as_real(audio.wav_to_stft(path)), torch.view_as_real(want))
    assert torch.equal(torch.view_as_real(audio.stft_from_waveform(loaded)), torch.view_as_real(want))


def test_codec_odd_sizes_against_oracle():
    """Ragged shapes: track lengths around the 2048-column block of the scan, nb_vec that is not a multiple of 4 (scalar stores)
    and one that does not divide the track (leading remainder dropped, functions.py:89-90)."""
    from musicgan_amd import audio
    from oracle import audio as OA
    rng = np.random.default_rng(11)
    for frames, nb in ((513, 512), (2049, 512), (2050, 512), (4097, 512), (6151, 512), (1000, 250), (777, 333), (130, 7)):
        x = ((rng.random((512, frames), dtype=np.float32) - 0.5) + 1j * (rng.random((512, frames), dtype=np.float32) - 0.5))
        x = x.astype(np.complex64)
        xd = torch.from_numpy(x).to(DEV)
        magn, phase = audio.stft_to_phase_magn(xd, nb_vec=nb)
        m_or, p_or = OA.stft_to_phase_magn(x, nb_vec=nb, lib="torch")  # th.abs / th.angle as the reference calls them (CPU)
        assert tuple(magn.shape) == m_or.shape == ((frames - 1) // nb, 512, nb), (frames, nb)
        assert np.array_equal(magn.cpu().numpy(), m_or), (frames, nb)
        assert np.array_equal(phase.cpu().numpy(), p_or), (frames, nb)


def test_codec_long_track_against_oracle():
    """2 000 frames of a tone + noise: the unwrap must reproduce the oracle's torch.cumsum-style (float64 accumulator) running
    sum.  Gated like the ragged sizes above: bit-identical to the oracle evaluated with torch's CPU abs / angle."""
    from musicgan_amd import audio
    from oracle import audio as OA
    rng = np.random.default_rng(3)
    wav = (rng.random(256 * 2000, dtype=np.float32) - 0.5)
    t = np.arange(wav.size) / 44100.0
    wav += (0.4 * np.sin(2 * np.pi * 880.0 * t)).astype(np.float32)
    c_ref = OA.stft(wav)
    xd = torch.from_numpy(c_ref).to(DEV)
    m_ref, p_ref = OA.stft_to_phase_magn(c_ref, lib="torch")
    magn, phase = audio.stft_to_phase_magn(xd)
    assert tuple(magn.shape) == m_ref.shape == (3, 512, 512)
    assert np.array_equal(magn.cpu().numpy(), m_ref) and np.array_equal(phase.cpu().numpy(), p_ref)


def test_drivers_end_to_end_tiny_corpus(tmp_path):
    """create_dataset -> train (a few iterations, batch 2) -> checkpoint -> generate, all through the reference-named
    drivers; checks shapes/ranges/finite-ness and the on-disk formats (create_dataset.py:51-64, utils.py:118-145)."""
    import musicgan_amd
    from musicgan_amd import audio
    from musicgan_amd.audio import wavio
    from musicgan_amd.networks import Generator
    rng = torch.Generator().manual_seed(5)
    wav_dir, data_dir, out_dir = tmp_path / "wav", tmp_path / "data", tmp_path / "out"
    wav_dir.mkdir()
    for i in range(2):
        wavio.save(str(wav_dir / f"s{i}.wav"), torch.rand(2, 256 * 1030, generator=rng) - 0.5, 44100)
    wavio.save(str(wav_dir / "short.wav"), torch.rand(1, 256 * 100, generator=rng) - 0.5, 44100)  # < 512 frames: skipped
    musicgan_amd.create_dataset(str(wav_dir / "*.wav"), str(data_dir))
    assert sorted(os.listdir(data_dir)) == sorted([f"magn_phase_{i}.pt" for i in range(4)] +  # 2 files x 2 samples
                                                  [f"magn_phase_f32.bin.{k}" for k in range(16)] +  # + the loader's side-car
                                                  ["magn_phase_f32.json"])                        #   (16 shard files)
    files = sorted(f for f in os.listdir(data_dir) if f.endswith(".pt"))
    sample = torch.load(str(data_dir / files[0]))
    assert sample.dtype == torch.float64 and tuple(sample.shape) == (2, 512, 512)
    assert float(sample.min()) >= -1.0 and float(sample.max()) <= 1.0
    assert len(audio.AudioDataset(str(data_dir))) == 4

    from musicgan_amd.train import train
    train("t", str(data_dir), str(out_dir), nb_epoch=3, batch_size=2, num_workers=0, max_iters=6, save_every=4)
    saved = sorted(f for f in os.listdir(out_dir) if f.endswith(".pt"))  # (+ 12 preview PNGs when matplotlib is installed)
    assert saved == ["disc_0.pt", "gen_0.pt", "optim_disc_0.pt", "optim_gen_0.pt", "train_state_0.pt"]
    st = torch.load(str(out_dir / "train_state_0.pt"))
    # the checkpoint is taken after the iteration's growth bookkeeping: 4 iterations x batch 2 = 8 samples seen
    assert st["iter_idx"] == 4 and st["level"] == 0 and st["grower"]["sample_idx"] == 8
    assert st["saver"] == {"calls": 4, "saves": 1} and st["epoch"] == 1 and st["epoch_pos"] == 2
    # checkpoint keys are the reference's (gen_{k}.pt loads into a reference Generator): utils.py:118-145
    gsd = torch.load(str(out_dir / "gen_0.pt"))
    assert "_Generator__gen_blocks.0.0.weight" in gsd and "_Generator__end_block.0.bias" in gsd
    osd = torch.load(str(out_dir / "optim_disc_0.pt"))
    assert set(osd.keys()) == {"state", "param_groups"} and "exp_avg_sq" in next(iter(osd["state"].values()))
    # resume: continues from iteration 4 with the saved weights / Adam state and runs two more iterations
    out2 = tmp_path / "out2"
    train("t2", str(data_dir), str(out2), nb_epoch=3, batch_size=2, num_workers=0, max_iters=6, save_every=100,
          resume_from=str(out_dir))
    # generate needs a level-7 checkpoint
    torch.manual_seed(0)
    g7 = Generator(8, end_layer=7)
    ck = str(tmp_path / "gen7.pt")
    torch.save(g7.state_dict(), ck)
    musicgan_amd.generate(str(tmp_path / "gen"), 8, ck, 1, 1)
    w, sr = wavio.load(str(tmp_path / "gen" / "sound_0.wav"))
    assert sr == 44100 and tuple(w.shape) == (1, 256 * 511) and bool(torch.isfinite(w).all())


def _tiny_dataset(tmp_path, n=6):
    """n stored samples (2, 512, 512) float64 in create_dataset's format, values as the codec would leave them."""
    data = tmp_path / "data"
    data.mkdir()
    rng = torch.Generator().manual_seed(17)
    for i in range(n):
        x = (torch.rand(2, 512, 512, generator=rng) * 2 - 1).double()  # float32 values widened, as create_dataset stores them
        torch.save(x, str(data / f"magn_phase_{i}.pt"))
    return data


def test_resume_is_bit_identical(tmp_path):
    """6 iterations straight == 4 iterations + resume + 2 (the reference can only save, utils.py:118-145):
    weights, Adam moments and step counts, Grower counters and the train state of the last checkpoint are identical bit for
    bit, across a growth before the interruption and another one after it; resuming IN PLACE continues the checkpoint
    numbering instead of overwriting `*_0.pt` (the files written before the interruption stay untouched)."""
    from musicgan_amd.train import train
    data = _tiny_dataset(tmp_path)
    kw = dict(nb_epoch=10, batch_size=2, num_workers=0, save_every=2, rand_channels=8,
              fadein_lengths=[1, 6, 6, 6, 6, 6, 6, 6], train_lengths=[5, 4, 100, 100, 100, 100, 100])
    torch.manual_seed(123)
    a = tmp_path / "straight"
    train("a", str(data), str(a), max_iters=6, **kw)
    torch.manual_seed(123)
    b = tmp_path / "interrupted"
    train("b", str(data), str(b), max_iters=4, **kw)
    before = {f: os.path.getmtime(str(b / f)) for f in os.listdir(b) if f.endswith(".pt")}
    assert sorted(before) == sorted(f"{s}_{k}.pt" for s in ("disc", "gen", "optim_disc", "optim_gen", "train_state")
                                    for k in (0, 1))
    torch.manual_seed(999)  # the resumed run must not depend on the process's RNG state
    train("b", str(data), str(b), max_iters=6, resume_from=str(b), **kw)
    assert all(os.path.getmtime(str(b / f)) == t for f, t in before.items()), "resume overwrote an earlier checkpoint"
    sa, sb = torch.load(str(a / "train_state_2.pt")), torch.load(str(b / "train_state_2.pt"))
    assert sa["level"] == sb["level"] == 2 and sa["iter_idx"] == sb["iter_idx"] == 6  # grew after iterations 3 and 5
    assert sa["grower"] == sb["grower"] and sa["saver"] == sb["saver"] == {"calls": 6, "saves": 3}
    assert (sa["epoch"], sa["epoch_pos"]) == (sb["epoch"], sb["epoch_pos"]) == (1, 3)  # end of the second pass over 6 samples
    assert all(torch.equal(x, y) for x, y in zip(sa["noise_rng"], sb["noise_rng"]))
    for net in ("gen", "disc"):
        wa, wb = torch.load(str(a / f"{net}_2.pt")), torch.load(str(b / f"{net}_2.pt"))
        assert list(wa.keys()) == list(wb.keys())
        for k in wa:
            assert torch.equal(wa[k], wb[k]), f"{net} {k} differs after resume"
        oa, ob = torch.load(str(a / f"optim_{net}_2.pt")), torch.load(str(b / f"optim_{net}_2.pt"))
        assert oa["state"].keys() == ob["state"].keys() and len(oa["param_groups"]) == len(ob["param_groups"]) == 3
        for ga, gb in zip(oa["param_groups"], ob["param_groups"]):  # (load_state_dict adds torch's default keys)
            assert all(ga[key] == gb[key] for key in ("lr", "betas", "eps", "params"))
        for i in oa["state"]:
            for key in ("step", "exp_avg", "exp_avg_sq"):
                assert torch.equal(oa["state"][i][key].cpu(), ob["state"][i][key].cpu()), f"optim_{net} state {i} {key}"


def test_training_loop_has_no_host_sync_between_metric_readbacks(tmp_path):
    """the reference blocks on 4-6 `.item()` calls per iteration (train.py:180-186,218-221).  Here a whole
    D step + G step + metric push (incl. both fused Adam steps and the device-side input transform) must not synchronise the host
    at all -- checked with PyTorch's sync debug mode, which raises on any blocking call -- and a metric window comes back with
    exactly one device-to-host copy."""
    from musicgan_amd.networks import Discriminator, Generator
    from musicgan_amd.optim import FusedAdam
    from musicgan_amd.train import MetricWindow
    from musicgan_amd.train_step import ProGANStepper
    from musicgan_amd.utils import Grower
    torch.manual_seed(0)
    gen, disc = Generator(8).to(DEV), Discriminator(7).to(DEV)
    for _ in range(2):
        gen.next_layer()
        disc.next_layer()
    og = FusedAdam(gen.parameters(), lr=1e-3, betas=(0.0, 0.9))
    od = FusedAdam(disc.parameters(), lr=1e-3, betas=(0.0, 0.9))
    noise = torch.Generator(device=DEV)
    noise.manual_seed(1)
    st = ProGANStepper(gen, disc, og, od, 8, noise=noise)
    grower = Grower(7, [1, 50, 50, 50, 50, 50, 50, 50], [1, 1, 1000, 1, 1, 1, 1])
    grower.grow(2)
    grower.grow(2)
    assert grower.curr_grow == 2
    metrics = MetricWindow(8, DEV)
    raw = torch.rand(4, 2, 512, 512, dtype=torch.float64).pin_memory()

    def iteration(i):
        x = grower.transform_batch(raw.to(DEV, non_blocking=True))
        d = st.d_step(x, grower.alpha)
        g = st.g_step(4, grower.alpha, DEV) if i % 5 == 0 else None
        metrics.push(d, g)
        grower.grow(4)

    iteration(0)  # warm-up: library load, workspaces, first allocations
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for i in range(5, 11):
            iteration(i)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with pytest.warns(UserWarning) as rec:
            metrics.flush()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert len([w for w in rec if "synchroniz" in str(w.message)]) == 1, [str(w.message) for w in rec]
    assert all(v == v for v in metrics.hist["disc_loss"][-7:]) and metrics.last["gen_loss"] == metrics.last["gen_loss"]


def test_create_dataset_sharded_equals_single_process(tmp_path, monkeypatch):
    """create_dataset under WORLD_SIZE=2 (files dealt round-robin to the ranks, no collective): rank 0 and rank 1 writing into one
    directory produce exactly the files of the single-process run -- same global `magn_phase_{idx}` numbering
    (create_dataset.py:32,64 of the reference: files in glob order, samples in time order), same contents; a file too short for
    one sample is skipped without consuming an index."""
    import musicgan_amd
    from musicgan_amd.audio import wavio
    rng = torch.Generator().manual_seed(9)
    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    lengths = [256 * 1030, 256 * 100, 256 * 520, 256 * 1600, 256 * 515]  # 2, 0 (skipped), 1, 3, 1 samples
    for i, n in enumerate(lengths):
        wavio.save(str(wav_dir / f"s{i}.wav"), torch.rand(2 if i % 2 else 1, n, generator=rng) - 0.5, 44100)
    single, sharded = tmp_path / "single", tmp_path / "sharded"
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    musicgan_amd.create_dataset(str(wav_dir / "*.wav"), str(single))
    for rank in (1, 0):
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "2")
        monkeypatch.setenv("LOCAL_RANK", "0")
        musicgan_amd.create_dataset(str(wav_dir / "*.wav"), str(sharded))
        if rank == 1:
            part = sorted(f for f in os.listdir(sharded) if f.endswith(".pt"))
            assert 0 < len(part) < 7  # rank 1 alone wrote only its files' samples
    pts = lambda d: sorted(f for f in os.listdir(d) if f.endswith(".pt"))  # (the single-process run also wrote the side-car)
    names = pts(single)
    assert names == pts(sharded) == sorted(f"magn_phase_{i}.pt" for i in range(7))
    for n in names:
        assert torch.equal(torch.load(str(single / n)), torch.load(str(sharded / n))), n


def test_packed_loader_serves_the_same_batches_and_the_same_training_run(tmp_path):
    """SURVEY 8(f) rank 1, second half (audio/dataset.py:14-44 + train.py:77-84,139): the float32 memory-mapped side-car and its
    double-buffered loader deliver exactly the samples `AudioDataset` + DataLoader deliver, in the same order, and a training run
    fed by it ends in bit-identical checkpoints."""
    from musicgan_amd import audio
    from musicgan_amd.train import ShardedShuffle, train
    data = _tiny_dataset(tmp_path, n=7)
    assert not audio.has_packed(str(data))
    assert audio.write_packed(str(data)) == 7 and audio.has_packed(str(data))
    ref_ds, ds = audio.AudioDataset(str(data)), audio.PackedAudioDataset(str(data))
    assert len(ds) == len(ref_ds) == 7
    for i in (0, 3, 6):
        assert ds[i].dtype == torch.float32 and torch.equal(ds[i].double(), ref_ds[i])
    sampler = ShardedShuffle(7, seed=5)
    sampler.set_epoch(2)
    order = list(iter(sampler))
    batches = [b.clone() for b in audio.PackedLoader(ds, 2, sampler, DEV)]
    assert len(batches) == 3  # drop_last
    for k, b in enumerate(batches):
        assert b.is_cuda and b.dtype == torch.float32
        want = torch.stack([ref_ds[i] for i in order[2 * k:2 * k + 2]])
        assert torch.equal(b.cpu().double(), want)
    # a sample file added behind the side-car's back invalidates it (the loader must never serve a stale index)
    torch.save(ref_ds[0], str(data / "magn_phase_7.pt"))
    assert not audio.has_packed(str(data))
    os.remove(str(data / "magn_phase_7.pt"))
    kw = dict(nb_epoch=4, batch_size=2, num_workers=0, save_every=5, rand_channels=8, max_iters=5)
    torch.manual_seed(321)
    train("p", str(data), str(tmp_path / "packed"), **kw)
    torch.manual_seed(321)
    train("r", str(data), str(tmp_path / "plain"), use_packed_loader=False, **kw)
    for f in ("gen_0.pt", "disc_0.pt"):
        a, b = torch.load(str(tmp_path / "packed" / f)), torch.load(str(tmp_path / "plain" / f))
        assert all(torch.equal(a[k], b[k]) for k in a), f


def test_packed_loader_with_a_gpu_that_lags_behind_the_host(tmp_path):
    """the consumer frees a loader slot with stream-level ordering only, so under graph replay (no host sync
    per iteration) the host runs many batches ahead of the GPU and an upload can still be QUEUED when the producer gets its pinned
    buffer back.  Every batch here has ~10 ms of GPU time queued in front of its use and the host never waits: the 13 batches must
    arrive with their own contents (the round-2 loader served later batches' samples / torn batches here)."""
    from musicgan_amd import audio
    data = _tiny_dataset(tmp_path, n=26)
    assert audio.write_packed(str(data)) == 26
    ds = audio.PackedAudioDataset(str(data))
    order = [int(i) for i in torch.randperm(26, generator=torch.Generator().manual_seed(3))]
    loader = audio.PackedLoader(ds, 2, order, DEV)
    got = torch.empty((13, 2, 2, 512, 512), device=DEV)
    for epoch in range(2):  # the second epoch re-uses the slots (and their pending uploads) of the first
        nb = 0
        for b, batch in enumerate(loader):
            torch.cuda._sleep(20_000_000)
            got[b].copy_(batch)
            nb += 1
        assert nb == 13
        torch.cuda.synchronize()
        for b in range(13):
            want = torch.stack([ds[i] for i in order[2 * b:2 * b + 2]])
            assert torch.equal(got[b].cpu(), want), f"epoch {epoch}, batch {b}"


def test_packed_loader_propagates_producer_failures_and_stops_cleanly(tmp_path):
    """an exception in the producer thread (I/O error on the memory map, HIP error) must surface in the
    consumer instead of leaving it blocked in `ready.get()` for ever; an early `break` must not leave the thread behind."""
    import threading
    from musicgan_amd import audio
    data = _tiny_dataset(tmp_path, n=12)
    audio.write_packed(str(data))
    ds = audio.PackedAudioDataset(str(data))
    real_gather, calls = ds.gather, []

    def failing_gather(indices, out):
        calls.append(list(indices))
        if len(calls) == 3:
            raise OSError("simulated read error on the side-car")
        return real_gather(indices, out)

    ds.gather = failing_gather
    loader = audio.PackedLoader(ds, 2, list(range(12)), DEV)
    seen = 0
    with pytest.raises(OSError, match="simulated read error"):
        for _ in loader:
            seen += 1
    assert seen == 2
    ds.gather = real_gather
    before = threading.active_count()
    for b, _ in enumerate(audio.PackedLoader(ds, 2, list(range(12)), DEV)):
        if b == 1:
            break  # producer is ahead of us, possibly blocked on the full queue
    import time
    t0 = time.time()
    while threading.active_count() > before and time.time() - t0 < 5:
        time.sleep(0.05)
    assert threading.active_count() <= before, "the loader's producer thread outlived the iteration"


def test_create_dataset_sidecar_is_rebuilt_not_reused(tmp_path):
    """create_dataset over a directory that already holds a side-car of OTHER data with the same file names must
    not leave that side-car valid; the side-car it writes (streamed in idx order, indexed in file-NAME order) serves exactly the
    .pt files, also past 10 samples where the two orders differ (magn_phase_10.pt < magn_phase_2.pt)."""
    import musicgan_amd
    from musicgan_amd import audio
    from musicgan_amd.audio import wavio
    rng = torch.Generator().manual_seed(21)
    wav_a, wav_b, out = tmp_path / "a", tmp_path / "b", tmp_path / "data"
    wav_a.mkdir()
    wav_b.mkdir()
    wavio.save(str(wav_a / "x.wav"), torch.rand(1, 256 * (512 * 12 + 40), generator=rng) - 0.5, 44100)  # 12 samples
    wavio.save(str(wav_b / "x.wav"), torch.rand(1, 256 * (512 * 12 + 40), generator=rng) - 0.5, 44100)  # 12 other samples
    musicgan_amd.create_dataset(str(wav_a / "*.wav"), str(out))
    assert audio.has_packed(str(out))
    first = audio.PackedAudioDataset(str(out))[3].clone()
    stats = {}
    musicgan_amd.create_dataset(str(wav_b / "*.wav"), str(out), stats=stats)
    assert stats["files"] == 1 and stats["samples"] == 12 and stats["pt_bytes"] == 12 * 4 * 2 ** 20
    assert audio.has_packed(str(out))
    ref_ds, ds = audio.AudioDataset(str(out)), audio.PackedAudioDataset(str(out))
    assert len(ds) == len(ref_ds) == 12 and not torch.equal(ds[3], first)
    for i in range(12):
        assert torch.equal(ds[i].double(), ref_ds[i]), i
    # multi-rank runs write no side-car -- and must not leave the old one behind
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "2", "0"
    try:
        musicgan_amd.create_dataset(str(wav_a / "*.wav"), str(out))
    finally:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            del os.environ[k]
    assert not audio.has_packed(str(out)) and not [f for f in os.listdir(str(out)) if f.startswith("magn_phase_f32")]


def test_device_crc32_of_the_float64_payload_equals_zlib():
    """mg_crc32_f64: the zip CRC-32 of a float32 sample's float64 widening, computed where the sample is -- against zlib.crc32 of
    x.double().tobytes() for the (2,512,512) sample shape (64 pieces of 64 KiB), the smallest and the largest supported sizes,
    and values that exercise every exponent range (denormals, zeros, negative zero, large)."""
    import zlib
    from musicgan_amd import ops
    g = torch.Generator().manual_seed(12)
    for n, per in ((3, 2 * 512 * 512), (5, 8192), (1, 8192 * 256), (2, 8192 * 4)):
        x = torch.randn(n, per, generator=g) * torch.pow(10.0, torch.randint(-42, 30, (n, per), generator=g).float())
        x[0, :4] = torch.tensor([0.0, -0.0, 1e-45, -3.4e38])
        got = ops.crc32_of_float64(x.to(DEV)).cpu().tolist()
        want = [zlib.crc32(x[i].double().numpy().tobytes()) & 0xFFFFFFFF for i in range(n)]
        assert got == want, (n, per)


def test_stft_with_other_window_and_hop_sizes():
    """wav_to_stft's nperseg / stride arguments (functions.py:38-41) beyond the drivers' 1024 / 256: the untuned radix-2 path against
    the oracle (float64 rfft) and torch.stft, incl. a hop that does not divide the window, stereo input and a pure tone landing on
    its bin; sizes that are not powers of two keep raising an AssertionError."""
    from musicgan_amd import audio
    from oracle import audio as OA
    rng = np.random.default_rng(8)
    wav = (rng.random(44100, dtype=np.float32) - 0.5)
    for n_fft, hop in ((64, 16), (256, 100), (512, 128), (2048, 512), (4096, 1000), (8192, 4096)):
        c = audio.stft_from_waveform(torch.from_numpy(wav), nperseg=n_fft, stride=hop).cpu().numpy()
        ref = OA.stft(wav, n_fft=n_fft, hop=hop)
        assert c.shape == ref.shape == (n_fft // 2, 1 + wav.size // hop)
        assert float(np.abs(c - ref).max()) <= 1e-5 * float(np.abs(ref).max()), (n_fft, hop)
        w = torch.hann_window(n_fft)
        ts = torch.stft(torch.from_numpy(wav), n_fft, hop_length=hop, win_length=n_fft, window=w, center=True, pad_mode="reflect",
                        normalized=False, onesided=True, return_complex=True) / w.pow(2).sum().sqrt()
        assert float((torch.from_numpy(c) - ts[:-1]).abs().max()) <= 1e-5 * float(ts.abs().max()), (n_fft, hop)
    stereo = np.stack([wav, wav[::-1].copy()])
    c = audio.stft_from_waveform(torch.from_numpy(stereo), nperseg=512, stride=64).cpu().numpy()
    ref = OA.stft(stereo, n_fft=512, hop=64)
    assert float(np.abs(c - ref).max()) <= 1e-5 * float(np.abs(ref).max())
    tone = np.sin(2 * np.pi * 37 * np.arange(8192) / 256.0).astype(np.float32)  # bin 37 of a 256-point window
    mag = np.abs(audio.stft_from_waveform(torch.from_numpy(tone), nperseg=256, stride=64).cpu().numpy())
    assert (mag[:, 4:-4].argmax(axis=0) == 37).all()
    with pytest.raises(AssertionError):
        audio.stft_from_waveform(torch.from_numpy(wav), nperseg=1000, stride=250)
