"""STFT / magnitude-phase codec with the reference's function signatures
(/root/reference/music_gan/audio/functions.py:26-139), evaluated by the HIP kernels `mg_stft_1024`, `mg_codec_fwd`,
`mg_codec_inv` on the current ROCm device.  Results come back as tensors on that device."""
from __future__ import annotations

from typing import Tuple

import torch as th

from . import constant, wavio
from .. import ops

_bark_cache = {}


def _device() -> th.device:
    if not th.cuda.is_available():
        from .._lib import MusicGanHipError
        raise MusicGanHipError("musicgan_amd.audio needs a ROCm GPU (no CPU fallback)")
    return th.device("cuda", th.cuda.current_device())


def _bark_vector(nb_freq: int, device) -> th.Tensor:
    """unit-norm 6*asinh(f/600) on linspace(20, 22050, nb_freq) (functions.py:29-35); 512 floats, built once per device."""
    key = (nb_freq, str(device))
    if key not in _bark_cache:
        scale = 6. * th.arcsinh(th.linspace(20., 44100 // 2, nb_freq) / 600.)
        _bark_cache[key] = (scale / scale.norm()).to(device).contiguous()
    return _bark_cache[key]


def bark_magn_scale(magn: th.Tensor, unscale: bool = False) -> th.Tensor:
    assert len(magn.size()) == 2, f"(STFT, TIME), actual = {magn.size()}"
    s = _bark_vector(magn.size()[0], magn.device)[:, None]
    return magn / s if unscale else magn * s


def _fast_stft(nperseg: int, stride: int) -> bool:
    """1024 / 256 (audio/constant.py; all the drivers use) runs the tuned kernel; any other power-of-two window the untuned one."""
    if nperseg == constant.N_FFT and stride == constant.STFT_STRIDE:
        return True
    assert 64 <= nperseg <= 8192 and nperseg & (nperseg - 1) == 0 and stride >= 1, \
        f"nperseg must be a power of two in [64, 8192] and stride >= 1, actual = ({nperseg}, {stride})"
    return False


def stft_from_waveform(raw_audio: th.Tensor, nperseg: int = constant.N_FFT, stride: int = constant.STFT_STRIDE) -> th.Tensor:
    """(channels, samples) or (samples,) -> complex64 (nperseg/2, 1 + samples//stride), Nyquist row dropped."""
    fast = _fast_stft(nperseg, stride)
    dev = raw_audio.device if raw_audio.is_cuda else _device()
    x = raw_audio.to(dev, th.float32)
    if x.dim() == 2 and x.shape[0] > 1:
        frames = x.t().contiguous()  # frames x channels: the mono mean (functions.py:49) is taken by the kernel
        return ops.stft_1024_pcm(frames) if fast else ops.stft_generic(ops.pcm_to_mono(frames), nperseg, stride)
    mono = x.reshape(-1).contiguous()
    return ops.stft_1024(mono) if fast else ops.stft_generic(mono, nperseg, stride)


def stft_from_pcm(pcm: th.Tensor, nperseg: int = constant.N_FFT, stride: int = constant.STFT_STRIDE) -> th.Tensor:
    """PCM frames (frames, channels) exactly as a WAV file stores them (wavio.load_pcm), on the device -> the same result as
    wav_to_stft on that file: normalisation to [-1, 1], mono mean and STFT in one launch (two for other window sizes)."""
    if _fast_stft(nperseg, stride):
        return ops.stft_1024_pcm(pcm)
    return ops.stft_generic(ops.pcm_to_mono(pcm), nperseg, stride)


def wav_to_stft(wav_p: str, nperseg: int = constant.N_FFT, stride: int = constant.STFT_STRIDE) -> th.Tensor:
    pcm, sr = wavio.load_pcm(wav_p)
    assert sr == constant.SAMPLE_RATE, \
        f"Audio sample rate must be {constant.SAMPLE_RATE}Hz, " \
        f"file \"{wav_p}\" is {sr}Hz"
    import numpy as np
    return stft_from_pcm(th.from_numpy(np.ascontiguousarray(pcm)).to(_device()), nperseg, stride)


def stft_to_phase_magn(complex_values: th.Tensor, nb_vec: int = constant.N_VEC) -> Tuple[th.Tensor, th.Tensor]:
    dev = complex_values.device if complex_values.is_cuda else _device()
    c = complex_values.to(dev, th.complex64)
    return ops.codec_fwd(c, _bark_vector(c.shape[0], dev), nb_vec)


def stft_to_stacked_phase_magn(complex_values: th.Tensor, nb_vec: int = constant.N_VEC) -> th.Tensor:
    """`th.stack(stft_to_phase_magn(c), dim=1)` -- the (S, 2, 512, nb_vec) tensor create_dataset.py:52-58 builds -- written once by
    the codec kernel instead of two images and a concatenation pass."""
    dev = complex_values.device if complex_values.is_cuda else _device()
    c = complex_values.to(dev, th.complex64)
    return ops.codec_fwd(c, _bark_vector(c.shape[0], dev), nb_vec, stacked=True)


def magn_phase_to_waveform(magn_phase: th.Tensor) -> th.Tensor:
    assert len(magn_phase.size()) == 4, \
        f"(N, 2, H, W), actual = {magn_phase.size()}"
    assert magn_phase.size()[1] == 2, \
        f"Channels must be equal to 2, actual = {magn_phase.size()[1]}"
    assert magn_phase.size()[2] == constant.N_FFT // 2, \
        f"Frequency size must be equal to {constant.N_FFT // 2}, " \
        f"actual = {magn_phase.size()[2]}"
    dev = magn_phase.device if magn_phase.is_cuda else _device()
    mp = magn_phase.to(dev, th.float32).contiguous()
    return ops.codec_inv(mp, _bark_vector(constant.N_FFT // 2, dev))


def magn_phase_to_wav(magn_phase: th.Tensor, wav_path: str, sample_rate: int):
    raw_audio = magn_phase_to_waveform(magn_phase)
    wavio.save(wav_path, raw_audio[None, :], sample_rate)
