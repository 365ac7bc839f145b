"""CPU oracle for the STFT / magnitude-phase codec path.  TEST INFRASTRUCTURE ONLY (see oracle/progan.py header).

numpy restatement of /root/reference/music_gan/audio/functions.py:
  :13-23   diff / unwrap  (== np.unwrap along time incl. the -pi -> +pi correction; the cumulative sum of the 2 pi
           adjustments accumulates in float64 and is rounded to float32 per frame, as torch.cumsum does on the CPU)
  :26-35   bark_magn_scale: s = 6*asinh(linspace(20, 22050, F)/600), normalised to unit L2 norm
  :38-62   wav_to_stft: mono mean, periodic Hann(1024), torchaudio.functional.spectrogram(power=None, normalized=True)
           (== centre/reflect-padded framed rFFT divided by sqrt(sum w^2)), Nyquist row dropped
  :65-94   stft_to_phase_magn: abs/angle, bark scale, unwrap, first difference, global min/max -> [-1,1],
           drop the leading T mod nb_vec frames, split into nb_vec-frame images
  :97-139  magn_phase_to_wav: inverse codec (its asymmetries kept), cumulative phase, polar -> complex,
           zero Nyquist row, inverse_spectrogram(normalized=True) (== torch.istft overlap-add / window envelope)

Third-party boundary: torchaudio (requirements.txt:5, unpinned, absent here).  Its two functional wrappers are restated
from their documented behaviour; the pin is torch.stft/torch.istft in the build container (tools/ref_loader.py stand-in)
plus the explicit DFT identity below -- "parity pinned on torch.stft, not on torchaudio" (DESIGN.md).
  :117-118 the inverse's cumulative phase is a Python loop of float32 adds: sequential float32, NOT torch.cumsum.
Golden vectors: tests/golden/audio_codec.npz (tools/gen_golden.py: audio_case) and, at BASELINE config 5's size (10-minute
track, 103 360 frames), tests/golden/audio_config5.npz (audio_config5_case).
"""
from __future__ import annotations

import numpy as np

N_FFT = 1024
N_VEC = 512
STFT_STRIDE = 256
SAMPLE_RATE = 44100


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(np.float32)


def stft(wav: np.ndarray, n_fft: int = N_FFT, hop: int = STFT_STRIDE) -> np.ndarray:
    """wav (C, L) or (L,) float32 -> complex64 (n_fft/2, 1 + L//hop).   X[k,t] = sum_n w[n] xpad[hop t + n] e^{-2 pi i k n/N} / sqrt(sum w^2)."""
    wav = np.asarray(wav, dtype=np.float32)
    mono = wav.mean(axis=0, dtype=np.float32) if wav.ndim == 2 else wav
    w = hann_periodic(n_fft)
    xp = np.pad(mono, (n_fft // 2, n_fft // 2), mode="reflect")
    t = 1 + mono.shape[0] // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(t)[:, None]
    frames = xp[idx] * w[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1).T  # (n_fft/2+1, T)
    spec = spec / np.sqrt(np.sum(w.astype(np.float64) ** 2))
    return spec[:-1, :].astype(np.complex64)


def bark_scale_vector(nb_freq: int = N_FFT // 2, lib: str = "numpy") -> np.ndarray:
    """functions.py:29-35.  lib="torch": the same three library calls (linspace, arcsinh, norm) made through torch on the CPU as the
    reference makes them -- numpy's differ from them by one ulp on part of the 512 entries, i.e. on whole rows of the magnitude."""
    if lib == "torch":
        import torch
        scale = 6. * torch.arcsinh(torch.linspace(20., 44100 // 2, nb_freq) / 600.)
        return (scale / scale.norm()).numpy()
    f = np.linspace(20.0, 44100 // 2, nb_freq, dtype=np.float32)
    s = (6.0 * np.arcsinh(f / np.float32(600.0))).astype(np.float32)
    return (s / np.sqrt(np.sum(s.astype(np.float64) ** 2)).astype(np.float32)).astype(np.float32)


def unwrap(phi: np.ndarray) -> np.ndarray:
    phi = np.asarray(phi, dtype=np.float32)
    dphi = np.zeros_like(phi)
    dphi[:, 1:] = phi[:, 1:] - phi[:, :-1]
    pi = np.float32(np.pi)
    two_pi = np.float32(2 * np.pi)
    dphi_m = np.mod(dphi + pi, two_pi) - pi
    dphi_m[(dphi_m == -pi) & (dphi > 0)] = pi
    adj = dphi_m - dphi
    adj[np.abs(dphi) < pi] = 0
    # functions.py:23 `phi_adj.cumsum(1)`: torch.cumsum on a CPU float32 tensor keeps its running sum in DOUBLE
    # (at::acc_type<float, false>) and rounds every output to float32 -- not a float32 running sum.  The two differ by
    # 5e-3 of the output range on a 10-minute track (unwrapped phase ~1e5 rad, ulp 2^-7), see DESIGN.md section 2.
    return phi + np.cumsum(adj.astype(np.float64), axis=1).astype(np.float32)


def _abs_angle(c: np.ndarray, lib: str):
    """functions.py:69-70 `th.abs` / `th.angle`: the two library calls of the codec.  lib="numpy" evaluates them with numpy's
    float32 hypot / atan2; lib="torch" with the very library the reference calls (torch on the CPU -- third-party, not reference
    code): numpy's and torch's atan2f differ by 1 ulp on ~40 % of the bins, and the exact running sum in `unwrap` turns that into
    a one-ulp(1e5 rad) change of ~0.1 % of a 10-minute track's phase image, so the oracle is bit-identical to the reference at
    that length only with lib="torch" (tests/test_oracle_golden.py::test_audio_oracle_config5_*)."""
    if isinstance(lib, tuple):  # (|X|, angle X) evaluated elsewhere, e.g. by the device's own math library: isolates the scan
        return np.asarray(lib[0], dtype=np.float32), np.asarray(lib[1], dtype=np.float32)
    if lib == "torch":
        # ATen runs abs / angle through SLEEF's vector routines on full vectors and through libm's scalar hypotf / atan2f on the
        # remainder elements at the end of every thread's chunk (TensorIterator's vectorized_loop), so the reference's own bits
        # depend on the thread count of the machine: 5 or 7 instead of 8 threads moves one element of a 512 x 6151 input by an
        # ulp, and the running sum carries it down the row.  The oracle evaluates the vector routine on EVERY element: one
        # thread, length padded to a multiple of 64 -- what the reference computes on all but a handful of elements, and what
        # the device restates (csrc/sleef_f32.h).
        import torch
        flat = np.ascontiguousarray(c, dtype=np.complex64).reshape(-1)
        pad = (-flat.size) % 64
        t = torch.from_numpy(np.concatenate([flat, np.ones(pad, np.complex64)]) if pad else flat)
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            m, a = torch.abs(t).numpy(), torch.angle(t).numpy()
        finally:
            torch.set_num_threads(nthr)
        return m[:flat.size].reshape(c.shape), a[:flat.size].reshape(c.shape)
    assert lib == "numpy", lib
    return np.abs(c).astype(np.float32), np.angle(c).astype(np.float32)


def stft_to_phase_magn(c: np.ndarray, nb_vec: int = N_VEC, lib="numpy"):
    magn, phase = _abs_angle(c, lib)
    magn = magn * bark_scale_vector(c.shape[0], "torch" if isinstance(lib, str) and lib == "torch" else "numpy")[:, None]
    phase = unwrap(phase)
    phase = phase[:, 1:] - phase[:, :-1]
    magn = magn[:, 1:]
    magn = (magn - magn.min()) / (magn.max() - magn.min())
    phase = (phase - phase.min()) / (phase.max() - phase.min())
    magn, phase = magn * np.float32(2.0) - np.float32(1.0), phase * np.float32(2.0) - np.float32(1.0)
    r = magn.shape[1] % nb_vec
    magn, phase = magn[:, r:], phase[:, r:]
    s = magn.shape[1] // nb_vec
    magn = magn.reshape(c.shape[0], s, nb_vec).transpose(1, 0, 2)
    phase = phase.reshape(c.shape[0], s, nb_vec).transpose(1, 0, 2)
    return np.ascontiguousarray(magn), np.ascontiguousarray(phase)


def istft(z: np.ndarray, n_fft: int = N_FFT, hop: int = STFT_STRIDE) -> np.ndarray:
    """complex (n_fft/2+1, T) -> float32 (hop*(T-1),), the inverse of the normalized centre-padded STFT."""
    w = hann_periodic(n_fft).astype(np.float64)
    z = z.astype(np.complex128) * np.sqrt(np.sum(w ** 2))
    t = z.shape[1]
    frames = np.fft.irfft(z.T, n=n_fft, axis=1) * w[None, :]
    total = n_fft + hop * (t - 1)
    y = np.zeros(total)
    env = np.zeros(total)
    for i in range(t):
        y[i * hop:i * hop + n_fft] += frames[i]
        env[i * hop:i * hop + n_fft] += w ** 2
    y = y[n_fft // 2: total - n_fft // 2]
    env = env[n_fft // 2: total - n_fft // 2]
    return (y / env).astype(np.float32)


def magn_phase_to_wav(mp: np.ndarray) -> np.ndarray:
    """(N, 2, 512, W) float32 -> (256*(N*W-1),) float32 waveform (functions.py:97-139 without the file write)."""
    assert mp.ndim == 4 and mp.shape[1] == 2 and mp.shape[2] == N_FFT // 2
    mp = np.asarray(mp, dtype=np.float32)
    flat = mp.transpose(1, 2, 0, 3).reshape(2, mp.shape[2], -1)
    magn, phase = flat[0].copy(), flat[1].copy()
    magn = (magn + np.float32(1.0)) / np.float32(2.0)
    magn = magn / bark_scale_vector(magn.shape[0])[:, None]
    magn = magn / (magn.max() - magn.min())
    phase = (phase + np.float32(1.0)) / np.float32(2.0) * np.float32(2.0) * np.float32(np.pi) - np.float32(np.pi)
    phase = np.cumsum(phase, axis=1, dtype=np.float32)
    phase = np.mod(phase, np.float32(2 * np.pi))
    real = magn * np.cos(phase)
    imag = magn * np.sin(phase)
    z = np.concatenate([real + 1j * imag, np.zeros((1, real.shape[1]), dtype=np.complex64)], axis=0)
    return istft(z)


def dft_bin(wav_mono: np.ndarray, k: int, t: int, n_fft: int = N_FFT, hop: int = STFT_STRIDE) -> complex:
    """Explicit fp64 DFT of one (bin, frame): the identity the STFT kernel's indexing is checked against."""
    w = hann_periodic(n_fft).astype(np.float64)
    xp = np.pad(np.asarray(wav_mono, dtype=np.float64), (n_fft // 2, n_fft // 2), mode="reflect")
    n = np.arange(n_fft)
    seg = xp[hop * t: hop * t + n_fft] * w
    return complex(np.sum(seg * np.exp(-2j * np.pi * k * n / n_fft)) / np.sqrt(np.sum(w ** 2)))
