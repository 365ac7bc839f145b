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


def test_codec_long_track_against_oracle():
    """2 000 frames: the sequential fp32 unwrap must track the oracle's torch.cumsum-style running sum."""
    from musicgan_amd import audio
    from oracle import audio as OA
    rng = np.random.default_rng(3)
    wav = (rng.random(256 * 2000, dtype=np.float32) - 0.5)
    t = np.arange(wav.size) / 44100.0
    wav += (0.4 * np.sin(2 * np.pi * 880.0 * t)).astype(np.float32)
    c_ref = OA.stft(wav)
    m_ref, p_ref = OA.stft_to_phase_magn(c_ref)
    magn, phase = audio.stft_to_phase_magn(torch.from_numpy(c_ref))
    assert tuple(magn.shape) == m_ref.shape == (3, 512, 512)
    assert float(np.abs(magn.cpu().numpy() - m_ref).max()) <= 5e-6
    assert float(np.abs(phase.cpu().numpy() - p_ref).max()) <= 5e-4


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
    files = sorted(os.listdir(data_dir))
    assert files == [f"magn_phase_{i}.pt" for i in range(4)]  # 2 files x 2 samples
    sample = torch.load(str(data_dir / files[0]))
    assert sample.dtype == torch.float64 and tuple(sample.shape) == (2, 512, 512)
    assert float(sample.min()) >= -1.0 and float(sample.max()) <= 1.0
    assert len(audio.AudioDataset(str(data_dir))) == 4

    from musicgan_amd.train import train
    train("t", str(data_dir), str(out_dir), nb_epoch=3, batch_size=2, num_workers=0, max_iters=6, save_every=4)
    saved = sorted(f for f in os.listdir(out_dir) if f.endswith(".pt"))  # (+ 12 preview PNGs when matplotlib is installed)
    assert saved == ["disc_0.pt", "gen_0.pt", "optim_disc_0.pt", "optim_gen_0.pt", "train_state_0.pt"]
    st = torch.load(str(out_dir / "train_state_0.pt"))
    assert st["iter_idx"] == 4 and st["level"] == 0 and st["grower"]["sample_idx"] == 6  # grow() runs after the save
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
